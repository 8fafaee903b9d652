// wavelet_table.cpp -- name -> (hlen, dec_lo, dec_hi, rec_lo, rec_hi) lookup.
//
// Replaces the reference's table scan (w_compute_filters_separable,
// pdwt/src/separable.cu:19-54) and its static data (pdwt/src/filters.cpp).  The
// data is generated from PyWavelets by tools/gen_filter_table.py.
#include "wavelet_table.hpp"

#include <strings.h>

namespace pdwt {

static const WaveletEntry kTable[] = {
#include "wavelet_table.inc"
};

static const int kCount = (int)(sizeof(kTable) / sizeof(kTable[0]));

int wavelet_count() { return kCount; }

const WaveletEntry* wavelet_at(int i) { return (i >= 0 && i < kCount) ? &kTable[i] : nullptr; }

const WaveletEntry* find_wavelet(const char* name) {
    if (!name) return nullptr;
    // Haar aliases of the reference (pdwt/src/separable.cu:24-28; "rbior1.1" sic), plus the
    // correctly spelled rbio1.1.  Unlike the reference they also work for the SWT.
    static const char* haar_alias[] = {"haar", "db1", "bior1.1", "rbior1.1", "rbio1.1"};
    for (const char* a : haar_alias)
        if (!strcasecmp(name, a)) name = "haar";
    for (int i = 0; i < kCount; i++)
        if (!strcasecmp(name, kTable[i].name)) return &kTable[i];
    return nullptr;
}

}  // namespace pdwt
