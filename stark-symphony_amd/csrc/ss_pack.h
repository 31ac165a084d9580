// Shared by the translation units of libss_verify.so: config checks, layouts, error text.  No HIP here.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "ss_abi.h"
#include "ss_layout.h"

namespace ss {

int set_err(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));  // thread-local text behind ss_last_error()

bool cfg_ok(const ss_stwo_cfg *c);
bool shape_ok(const ss_s101_shape *sh);
StwoLayout lay_of(const ss_stwo_cfg *c, size_t n, bool minimal = false);  // minimal: the layout behind minimal records (ss_minimal.h)

}  // namespace ss
