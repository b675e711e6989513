// common.hpp -- shared declarations of the engine's translation units.
#pragma once
#include <string>
#include "../../include/spasm_amd.h"

#define SPASM_API __attribute__((visibility("default")))

// error text of the calling thread (returned by spasm_amd_last_error); also logged
void spasm_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void spasm_clear_error();
// progress text: to `logcallback` when set, else to stderr (reference src/SpaSM.jl:34-46, :838-858)
void spasm_logf(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// rank certificates (abi.cpp): the Fiat-Shamir challenge x of a certificate -- r balanced residues drawn from SHA-256(hash, prime,
// r, rows, columns) -- and the host-side check  sum_k y[k] * A[i[k]] restricted to the columns j[]  ==  x
extern "C" void spasm_cert_challenge(const uint8_t *hash, i64 prime, int r, const int *i, const int *j, spasm_ZZp *x);
