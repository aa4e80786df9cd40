// brx_internal.h -- what brx_node.cpp needs from brx_api.cpp beyond the public C ABI (not part of include/brx.h).
#pragma once
#include <stdint.h>

#include "../../include/brx.h"

int brx_fail(int code, const char *what);        // sets brx_last_error() of the calling thread, returns `code`
int brx_ctx_device(const brx_ctx *c);             // HIP device index of a context
unsigned brx_ctx_max_grid(const brx_ctx *c);      // regular-kernel waves the GPU runs at a time (16 per CU)
void brx_launch_ragged_copy(const void *src, const uint64_t *src_off, const uint64_t *len, void *dst, const uint64_t *dst_off,
                            const uint64_t *part_off, uint32_t n, uint64_t total, void *hip_stream); // brx_util.hip
