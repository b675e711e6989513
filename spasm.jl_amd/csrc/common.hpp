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
