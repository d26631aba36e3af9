// rd_ljpeg.h -- lossless JPEG (ITU-T T.81 process 14, SOF3: Huffman, predictive) decoder for the ingest side.
//
// The reference decodes RAW files with the third-party `rawloader` crate (src/raw/loader.rs:50-54; not vendored, so no
// parity claim against it).  Compressed DNGs -- what Adobe's converter and most cameras that write DNG produce -- keep
// their CFA samples as lossless-JPEG tiles (TIFF Compression = 7); this is the decoder raweditor_amd/ingest.py uses for
// them.  Lossless JPEG is exact by construction: the decoded samples are the encoder's input, bit for bit, which is what
// tests/test_ingest_catalog_cpu.py checks with its own encoder.  Host code only (no device work).
//
// Supported: 1-4 components (all in one scan, interleaved), precision 2-16, predictors 1-7, point transform, restart
// intervals, up to four Huffman tables, 0xFF00 byte stuffing.  Not supported (error): arithmetic coding, hierarchical
// mode, several scans.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace rd_ljpeg {

struct huff {
    // canonical JPEG Huffman table (T.81 Annex C / F.2.2.3): codes of length l occupy [mincode[l], maxcode[l]]
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
    uint8_t look_len[256], look_val[256];     // 8-bit lookahead: code length (0 = longer than 8) and symbol
};

struct bits {
    const uint8_t *p, *end;
    uint64_t acc = 0;
    int n = 0;
    int64_t injected = 0;                    // zero bits appended after a marker or the end of the input (64-bit: a
                                             // damaged stream may ask for more than 2^31 of them before a row check)
    bool hit_marker = false;
    void fill()
    {
        while (n <= 56) {
            uint32_t b = 0;
            bool real = false;
            if (!hit_marker && p < end) {
                if (*p != 0xff) { b = *p++; real = true; }
                else if (p + 1 < end && p[1] == 0x00) { b = 0xff; p += 2; real = true; }      // stuffed zero after a data 0xFF
                else hit_marker = true;                                                        // a marker: left for the caller
            }
            if (!real) injected += 8;                                                          // zeros from here on
            acc |= (uint64_t)b << (56 - n);
            n += 8;
        }
    }
    uint32_t peek(int k) { if (n < k) fill(); return (uint32_t)(acc >> (64 - k)); }
    void drop(int k) { acc <<= k; n -= k; }
    uint32_t get(int k) { if (!k) return 0; const uint32_t v = peek(k); drop(k); return v; }
    void reset() { acc = 0; n = 0; injected = 0; hit_marker = false; }
    bool overran() const { return injected > n; }     // bits that were never in the stream have been consumed
};

inline bool build(huff &h, const uint8_t counts[16], const uint8_t *vals, int nvals)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h.valptr[l] = k;
        h.mincode[l] = code;
        code += counts[l - 1];
        k += counts[l - 1];
        h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
        if (code > (1 << l)) return false;                                    // over-subscribed
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    if (k != nvals || k > 256) return false;
    memcpy(h.vals, vals, (size_t)k);
    memset(h.look_len, 0, sizeof h.look_len);
    for (int l = 1; l <= 8; ++l)
        for (int c = h.mincode[l]; h.maxcode[l] >= 0 && c <= h.maxcode[l]; ++c) {
            const uint8_t v = h.vals[h.valptr[l] + c - h.mincode[l]];
            for (int pad = 0; pad < (1 << (8 - l)); ++pad) {
                h.look_len[(c << (8 - l)) | pad] = (uint8_t)l;
                h.look_val[(c << (8 - l)) | pad] = v;
            }
        }
    h.present = true;
    return true;
}

inline int decode_symbol(const huff &h, bits &b)
{
    const uint32_t look = b.peek(16);
    const uint32_t top = look >> 8;
    if (h.look_len[top]) { b.drop(h.look_len[top]); return h.look_val[top]; }
    for (int l = 9; l <= 16; ++l) {
        const int32_t c = (int32_t)(look >> (16 - l));
        if (h.maxcode[l] >= 0 && c <= h.maxcode[l] && c >= h.mincode[l]) { b.drop(l); return h.vals[h.valptr[l] + c - h.mincode[l]]; }
    }
    return -1;
}

inline int decode_diff(const huff &h, bits &b, bool &ok)
{
    const int s = decode_symbol(h, b);
    if (s < 0 || s > 16) { ok = false; return 0; }
    if (s == 0) return 0;
    if (s == 16) return 32768;                                                // no extra bits (T.81 H.1.2.2)
    int v = (int)b.get(s);
    if (v < (1 << (s - 1))) v -= (1 << s) - 1;                                // EXTEND
    return v;
}

enum { OK = 0, ERR_FORMAT = -1, ERR_UNSUPPORTED = -2, ERR_TRUNCATED = -3, ERR_SIZE = -4 };

// Decodes one lossless-JPEG stream.  dst receives height * width * ncomp samples, row-major, components interleaved;
// dst_cap is its capacity in samples.  *w, *h, *nc, *prec describe the frame.
inline int decode(const uint8_t *src, size_t len, uint16_t *dst, size_t dst_cap, uint32_t *w, uint32_t *h, uint32_t *nc, uint32_t *prec)
{
    if (len < 4 || src[0] != 0xff || src[1] != 0xd8) return ERR_FORMAT;
    huff tabs[4];
    uint32_t W = 0, H = 0, N = 0, P = 0, restart = 0;
    int comp_tab[4] = { 0, 0, 0, 0 }, comp_id[4] = { 0, 0, 0, 0 };
    size_t pos = 2;
    bool have_sof = false;
    for (;;) {
        if (pos + 4 > len) return ERR_TRUNCATED;
        if (src[pos] != 0xff) return ERR_FORMAT;
        const uint8_t m = src[pos + 1];
        if (m == 0xff) { ++pos; continue; }                                    // fill bytes
        const size_t seg = ((size_t)src[pos + 2] << 8) | src[pos + 3];
        if (seg < 2 || pos + 2 + seg > len) return ERR_TRUNCATED;
        const uint8_t *d = src + pos + 4;
        const size_t dl = seg - 2;
        if (m == 0xc4) {                                                       // DHT
            size_t o = 0;
            while (o + 17 <= dl) {
                const int tc = d[o] >> 4, th = d[o] & 15;
                int total = 0;
                for (int i = 0; i < 16; ++i) total += d[o + 1 + i];
                if (tc != 0 || th > 3 || o + 17 + (size_t)total > dl) return ERR_FORMAT;
                if (!build(tabs[th], d + o + 1, d + o + 17, total)) return ERR_FORMAT;
                o += 17 + (size_t)total;
            }
        } else if (m == 0xc3) {                                                // SOF3
            if (dl < 6) return ERR_FORMAT;
            P = d[0]; H = ((uint32_t)d[1] << 8) | d[2]; W = ((uint32_t)d[3] << 8) | d[4]; N = d[5];
            if (P < 2 || P > 16 || !W || !H || N < 1 || N > 4 || dl < 6 + 3 * (size_t)N) return ERR_UNSUPPORTED;
            for (uint32_t c = 0; c < N; ++c) {
                comp_id[c] = d[6 + 3 * c];
                if (d[7 + 3 * c] != 0x11) return ERR_UNSUPPORTED;             // sub-sampling makes no sense for CFA data
            }
            have_sof = true;
        } else if (m == 0xdd) {                                                // DRI
            if (dl < 2) return ERR_FORMAT;
            restart = ((uint32_t)d[0] << 8) | d[1];
        } else if (m == 0xda) {                                                // SOS
            if (!have_sof || dl < 1 || d[0] != N || dl < 1 + 2 * (size_t)N + 3) return ERR_UNSUPPORTED;
            for (uint32_t c = 0; c < N; ++c) {
                if (d[1 + 2 * c] != comp_id[c]) return ERR_UNSUPPORTED;
                comp_tab[c] = d[2 + 2 * c] >> 4;
                if (comp_tab[c] > 3 || !tabs[comp_tab[c]].present) return ERR_FORMAT;
            }
            const int pred = d[1 + 2 * N], pt = d[3 + 2 * N] & 15;
            if (pred < 1 || pred > 7 || (uint32_t)pt >= P) return ERR_UNSUPPORTED;
            // a restart interval is a whole number of lines (T.81 H.1.2.1 as DNG writers use it): anything else would be
            // predicted wrongly below (first sample after RSTn from Ra instead of 2^(P-Pt-1)) without any error
            if (restart && restart % W) return ERR_UNSUPPORTED;
            const size_t total = (size_t)W * H * N;
            if (w) *w = W;                                                     // reported BEFORE the capacity check, so a caller
            if (h) *h = H;                                                     // can ask for the frame size with capacity 0
            if (nc) *nc = N;
            if (prec) *prec = P;
            if (total > dst_cap) return ERR_SIZE;
            bits b;
            b.p = src + pos + 2 + seg;
            b.end = src + len;
            const int init = 1 << (P - pt - 1);
            const size_t stride = (size_t)W * N;
            uint32_t until_restart = restart;
            bool fresh = true;                                                 // at the start of the scan / after a restart marker
            bool ok = true;
            for (uint32_t y = 0; y < H; ++y) {
                uint16_t *row = dst + (size_t)y * stride;
                const uint16_t *up = y ? row - stride : nullptr;
                for (uint32_t x = 0; x < W; ++x) {
                    if (restart && until_restart == 0) {                       // expect RSTn here
                        if (b.overran()) return ERR_TRUNCATED;
                        b.reset();
                        while (b.p + 1 < b.end && !(b.p[0] == 0xff && b.p[1] >= 0xd0 && b.p[1] <= 0xd7)) ++b.p;
                        if (b.p + 1 >= b.end) return ERR_TRUNCATED;
                        b.p += 2;
                        until_restart = restart;
                        fresh = true;
                    }
                    // the first line of a restart interval is predicted like the first line of the image (T.81 H.1.2.1; an
                    // interval is a whole number of lines, so a restart can only fall at x == 0)
                    const bool first_line = fresh || !up;
                    for (uint32_t c = 0; c < N; ++c) {
                        int px;
                        if (first_line && x == 0) px = init;
                        else if (first_line) px = row[(size_t)(x - 1) * N + c];                       // Ra
                        else if (x == 0) px = up[c];                                                   // Rb
                        else {
                            const int ra = row[(size_t)(x - 1) * N + c], rb = up[(size_t)x * N + c], rc = up[(size_t)(x - 1) * N + c];
                            switch (pred) {
                            case 1: px = ra; break;
                            case 2: px = rb; break;
                            case 3: px = rc; break;
                            case 4: px = ra + rb - rc; break;
                            case 5: px = ra + ((rb - rc) >> 1); break;
                            case 6: px = rb + ((ra - rc) >> 1); break;
                            default: px = (ra + rb) >> 1; break;
                            }
                        }
                        const int diff = decode_diff(tabs[comp_tab[c]], b, ok);
                        if (!ok) return ERR_FORMAT;
                        row[(size_t)x * N + c] = (uint16_t)((px + diff) & 0xffff);
                    }
                    if (restart) --until_restart;
                    if (x == W - 1) fresh = false;                             // "fresh" lasts for one line
                }
                // a truncated stream is refused after the row in which it ran dry, not decoded to the end of the declared
                // frame out of injected zero bits
                if (b.overran()) return ERR_TRUNCATED;
            }
            if (b.overran()) return ERR_TRUNCATED;                             // the entropy-coded segment ended before the image did
            if (pt) for (size_t i = 0; i < total; ++i) dst[i] = (uint16_t)(dst[i] << pt);
            return OK;
        } else if (m == 0xd9) {
            return ERR_FORMAT;                                                 // EOI before any scan
        } else if ((m >= 0xc0 && m <= 0xcf) && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return ERR_UNSUPPORTED;                                            // another SOF: not lossless Huffman
        }
        pos += 2 + seg;
    }
}

}  // namespace rd_ljpeg
