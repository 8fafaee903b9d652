// wavelet_table.hpp -- built-in wavelet filter banks (72 names of the reference).
#pragma once

namespace pdwt {

struct WaveletEntry {
    const char* name;
    int hlen;
    double dec_lo[40];
    double dec_hi[40];
    double rec_lo[40];
    double rec_hi[40];
};

int wavelet_count();
const WaveletEntry* wavelet_at(int i);
// case-insensitive; resolves the Haar aliases; nullptr when unknown
const WaveletEntry* find_wavelet(const char* name);

}  // namespace pdwt
