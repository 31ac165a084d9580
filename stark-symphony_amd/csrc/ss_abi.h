// The three public headers of the C ABI (include/): what every translation unit behind them compiles against.
#pragma once
#include "../../include/ss_verify.h"
#include "../../include/ss_verify_forms.h"
#include "../../include/ss_verify_test.h"
