# copy the summaries of gpurun_out/final (tools/final_profile.sh) into profiles/ under the given tag: bash tools/collect_profiles.sh r01_final
tag=${1:-r01_final}
cd "$(dirname "$0")/.."
python tools/pmc_summary.py gpurun_out/final profiles/${tag}_traffic.json > /dev/null
cp "$(ls -t gpurun_out/final/trace/*/*_kernel_stats.csv | head -1)" profiles/${tag}_kernel_stats.csv
cp gpurun_out/final/bench_full.json profiles/${tag}_bench.json
cp "$(ls -t gpurun_out/final/calib_FETCH_SIZE/*/*_counter_collection.csv | head -1)" profiles/${tag}_calibration_FETCH_SIZE.csv
cp "$(ls -t gpurun_out/final/calib_WRITE_SIZE/*/*_counter_collection.csv | head -1)" profiles/${tag}_calibration_WRITE_SIZE.csv
cp gpurun_out/final/shard_time.txt profiles/${tag}_shard_time.txt
[ -f gpurun_out/final/ablate_scatter.txt ] && cp gpurun_out/final/ablate_scatter.txt profiles/${tag}_ablations_scatter.txt
[ -f gpurun_out/final/ablate_combine.txt ] && cp gpurun_out/final/ablate_combine.txt profiles/${tag}_ablations_combine.txt
for k in 'k_wstream<10' 'k_wstream<11' 'k_wstream<12' 'k_wplan' 'k_wlevel_wave' 'k_wlevel_wg' 'k_scatter<13'; do echo "== $k (mean per dispatch)"; python tools/pmc_show.py gpurun_out/pmc_final "$k"; done > profiles/${tag}_instruction_mix.txt
ls -la profiles | grep ${tag}
