"""Summarise the rocprofv3 PMC passes of tools/final_profile.sh into profiles/<tag>_traffic.json.

    python tools/pmc_summary.py gpurun_out/final profiles/r01_final_traffic.json

Reads <dir>/pmc_FETCH_SIZE, <dir>/pmc_WRITE_SIZE, <dir>/pmc_TCC_HIT_sum_TCC_MISS_sum (one counter set per pass, as the
MI355X guide prescribes) and writes, per kernel of the Schur round, the mean per dispatch of
  fetch_bytes = FETCH_SIZE [KiB] * 1024,  write_bytes = WRITE_SIZE [KiB] * 1024,  l2_hit_rate = HIT / (HIT + MISS).
The calibration factors of tools/fetch_calib.hip for these access shapes are 1.000 (see profiles/README.md), so no
correction is applied to the scatter kernels.  Kernel names are cut at the '(' of the argument list.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(dirname):
    """{counter: {kernel: [values]}} over every dispatch of the pass"""
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                out[row["Counter_Name"]][name].append(float(row["Counter_Value"]))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    fetch = per_kernel(os.path.join(src, "pmc_FETCH_SIZE")).get("FETCH_SIZE", {})
    write = per_kernel(os.path.join(src, "pmc_WRITE_SIZE")).get("WRITE_SIZE", {})
    tcc = per_kernel(os.path.join(src, "pmc_TCC_HIT_sum_TCC_MISS_sum"))
    mean = lambda v: sum(v) / len(v) if v else 0.0
    kernels = {}
    for k in sorted(fetch):
        if not k.startswith(("k_scatter", "k_combine", "k_bin", "k_solve", "k_wstream", "k_wplan", "k_stream_fix", "k_wlevel", "k_wbuild")):
            continue
        hit, miss = mean(tcc.get("TCC_HIT_sum", {}).get(k, [])), mean(tcc.get("TCC_MISS_sum", {}).get(k, []))
        # per Schur step: a kernel launched several times per step (the level kernels of the W build: one launch per level) is
        # summed over its launches and divided by the number of steps of the run -- the builds (launches of k_wbuild_reset,
        # the first one at plan creation); the kernels of the plan and the scatter are launched once per step: their mean per dispatch
        multi = k.startswith(("k_wlevel", "k_wbuild"))
        fsteps = len(fetch.get("k_wbuild_reset", [])) if multi else len(fetch[k])
        wsteps = len(write.get("k_wbuild_reset", [])) if multi else len(write.get(k, []))
        kernels[k] = {
            "dispatches": len(fetch[k]),
            "fetch_bytes_per_step": int(sum(fetch[k]) * 1024 / fsteps) if fsteps else None,
            "write_bytes_per_step": int(sum(write.get(k, [])) * 1024 / wsteps) if wsteps else None,
            "fetch_bytes": int(mean(fetch[k]) * 1024),
            "write_bytes": int(mean(write.get(k, [])) * 1024),
            "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss > 0 else None,
        }
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate passes "
            "(python3 bench.py --steps 3 --warmup 1), mean per dispatch, KiB * 1024. Calibrated with tools/fetch_calib.hip "
            "on the same box: FETCH_SIZE = 1.000 x bytes for 8-byte-per-lane reads in 64-byte segments (the pivot-row "
            "access shape of k_scatter), 0.500 x for a coalesced 8-byte-per-lane stream; WRITE_SIZE = 1.000 x for "
            "compacted 8-byte stores.")
    json.dump({"_note": note, "kernels": kernels}, open(dst, "w"), indent=1)
    for k, v in kernels.items():
        print(k, v)


if __name__ == "__main__":
    main()
