// tuning.hpp -- the dispatch thresholds of tuning_gfx950.inc as constants: pdwt::tune::key
#pragma once

namespace pdwt {
namespace tune {
#define PDWT_TUNE(key, value, evidence, measured, what) constexpr long long key = value;
#include "tuning_gfx950.inc"
#undef PDWT_TUNE

struct Row {
    const char *key, *evidence, *measured, *what;
    long long value;
};
// the same rows as text (pdwt_tuning_table, tools/tuning_table.py)
static const Row kRows[] = {
#define PDWT_TUNE(key, value, evidence, measured, what) {#key, evidence, measured, what, value},
#include "tuning_gfx950.inc"
#undef PDWT_TUNE
};
constexpr int kRowCount = sizeof(kRows) / sizeof(kRows[0]);
}  // namespace tune
}  // namespace pdwt
