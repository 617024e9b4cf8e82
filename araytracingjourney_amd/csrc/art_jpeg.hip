// art_jpeg.hip -- JPEG (ITU T.81: baseline / extended sequential and progressive DCT, Huffman, 8 bit) for the GLB reader: the reference reads its models through
// the `gltf` crate's import(), whose images are decoded by the `image` crate (PNG and JPEG; model_reader/gltf_model_reader.rs:55-70).
// Grey and YCbCr, sampling factors 1 or 2 per axis, restart intervals; lossless / hierarchical / arithmetic / 12-bit streams are errors.
// Chroma is upsampled with the triangle filter of libjpeg ("fancy upsampling", also what the crate's decoder does for h2v1 / h2v2) and
// converted with the JFIF matrix; the inverse DCT is the separable float one, so values can differ from another decoder's by an LSB or two.
// Host code only.
// (no HIP headers: this file and art_glb.hip also build with g++ under AddressSanitizer, tests/test_glb.py)
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace art { bool decode_jpeg(const uint8_t *data, size_t n, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, int &channels, std::string &err); } // also declared in art_internal.h

namespace art {
namespace {

struct BitReader {
    const uint8_t *p, *e; uint32_t acc = 0; int n = 0; bool hit_marker = false;
    int bit() {
        if (n == 0) {
            uint8_t b = p < e ? *p++ : 0;
            if (b == 0xFF) { uint8_t b2 = p < e ? *p : 0; if (b2 == 0) p++; else { hit_marker = true; b = 0; p--; } } // a marker inside entropy data: feed zeros
            acc = b; n = 8;
        }
        n--;
        return (int)((acc >> n) & 1u);
    }
    int bits(int k) { int v = 0; while (k--) v = (v << 1) | bit(); return v; }
    void reset() { n = 0; hit_marker = false; }
};
struct Huff { int mincode[17], maxcode[18], valptr[17]; uint8_t vals[256]; bool ok = false;
    void build(const uint8_t *counts, const uint8_t *symbols) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) { valptr[l] = k; mincode[l] = code; code += counts[l - 1]; k += counts[l - 1]; maxcode[l] = counts[l - 1] ? code - 1 : -1; code <<= 1; }
        maxcode[17] = 0x7FFFFFFF;
        std::memcpy(vals, symbols, (size_t)k);
        ok = true;
    }
    int decode(BitReader &br) const {
        int code = 0;
        for (int l = 1; l <= 16; l++) { code = (code << 1) | br.bit(); if (maxcode[l] >= 0 && code <= maxcode[l] && code >= mincode[l]) return vals[valptr[l] + code - mincode[l]]; }
        return -1;
    }
};
inline int extend(int v, int t) { return t && v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const float *in, uint8_t *out, int stride) { // separable, out = clamp(round(x + 128))
    static float c[8][8]; static bool init = false;
    if (!init) { for (int u = 0; u < 8; u++) for (int x = 0; x < 8; x++) c[u][x] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f); init = true; }
    float tmp[64];
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { float s = 0; for (int u = 0; u < 8; u++) s += c[u][x] * in[y * 8 + u]; tmp[y * 8 + x] = s; }
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
        float s = 0; for (int v = 0; v < 8; v++) s += c[v][y] * tmp[v * 8 + x];
        int q = (int)std::lrintf(s + 128.0f);
        out[y * stride + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
}

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, w = 0, hgt = 0, bw = 0, bh = 0, cbw = 0, cbh = 0;
              std::vector<uint8_t> plane; std::vector<int16_t> coef; }; // bw x bh: blocks incl. MCU padding; cbw x cbh: blocks that cover the component

struct Decoder {
    uint16_t qt[4][64] = {}; bool have_qt[4] = {};
    Huff hdc[4], hac[4];
    std::vector<Comp> comps;
    int W = 0, H = 0, restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false;
    std::string err;

    bool fail(const char *m) { err = m; return false; }
    void skip_to_restart(BitReader &br) { // RSTn: byte-align, skip the marker
        br.reset();
        while (br.p + 1 < br.e && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
        if (br.p + 1 < br.e) br.p += 2;
    }
    // one block of a baseline scan or of a progressive DC / AC first / refinement pass; coefficients in natural order
    bool block(BitReader &br, Comp &c, int16_t *q, int ss, int se, int ah, int al, int &eobrun) {
        if (!progressive || (ss == 0 && ah == 0)) { // DC, first pass (baseline: the whole block follows)
            int t = hdc[c.td].decode(br);
            if (t < 0 || t > 11) return fail("corrupt JPEG entropy data (DC)");
            c.pred += extend(br.bits(t), t);
            q[0] = (int16_t)(c.pred * (1 << al));
            if (progressive) return true;
        } else if (ss == 0) { if (br.bit()) q[0] = (int16_t)(q[0] | (1 << al)); return true; } // DC refinement
        if (!progressive) { ss = 1; se = 63; }
        if (ah == 0) { // AC first pass (and the AC part of a baseline block)
            if (eobrun > 0) { eobrun--; return true; }
            for (int k = ss; k <= se;) {
                int rs = hac[c.ta].decode(br);
                if (rs < 0) return fail("corrupt JPEG entropy data (AC)");
                int r = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                    if (r == 15) { k += 16; continue; }
                    if (progressive) { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); }
                    break;
                }
                k += r;
                if (k > se) return fail("corrupt JPEG entropy data (run past the band)");
                q[kZigzag[k]] = (int16_t)(extend(br.bits(sz), sz) * (1 << al));
                k++;
            }
            return true;
        }
        // AC refinement (T.81 G.1.2.3): correction bits for the non-zero history, new +-1 coefficients placed after `r` zero-history positions
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se;) {
                int rs = hac[c.ta].decode(br);
                if (rs < 0) return fail("corrupt JPEG entropy data (AC refinement)");
                int r = rs >> 4, sz = rs & 15, val = 0;
                if (sz == 0) { if (r != 15) { eobrun = (1 << r); if (r) eobrun += br.bits(r); break; } } // EOBn: this block is the first of the run
                else if (sz == 1) val = br.bit() ? p1 : m1;
                else return fail("corrupt JPEG entropy data (refinement magnitude)");
                for (; k <= se; k++) {
                    int16_t &cf = q[kZigzag[k]];
                    if (cf != 0) { if (br.bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1); }
                    else { if (r == 0) { if (val) cf = (int16_t)val; k++; break; } r--; }
                }
            }
        }
        if (eobrun > 0) { // the rest of the band: only correction bits
            for (; k <= se; k++) { int16_t &cf = q[kZigzag[k]]; if (cf != 0 && br.bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1); }
            eobrun--;
        }
        return true;
    }
    bool scan(const uint8_t *s, size_t sl, const uint8_t *data, const uint8_t *end) {
        int ns = s[0];
        if (ns < 1 || ns > (int)comps.size() || sl < (size_t)(1 + 2 * ns + 3)) return fail("bad JPEG scan header");
        Comp *sc[3];
        for (int k = 0; k < ns; k++) {
            int cid = s[1 + 2 * k]; sc[k] = nullptr;
            for (auto &cc : comps) if (cc.id == cid) sc[k] = &cc;
            if (!sc[k]) return fail("JPEG scan names an unknown component");
            sc[k]->td = s[2 + 2 * k] >> 4; sc[k]->ta = s[2 + 2 * k] & 15;
            if (sc[k]->td > 3 || sc[k]->ta > 3) return fail("bad JPEG table selector");
        }
        int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
        if (!progressive) { if (ns != (int)comps.size()) return fail("JPEG with a partial scan (not baseline interleaved)"); ss = 0; se = 63; ah = al = 0; }
        else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) return fail("bad progressive JPEG scan parameters");
        for (int k = 0; k < ns; k++) {
            if ((ss == 0 && ah == 0 && !hdc[sc[k]->td].ok) || ((se > 0 || !progressive) && !hac[sc[k]->ta].ok) || !have_qt[sc[k]->tq]) return fail("JPEG scan uses a table that was not defined");
            sc[k]->pred = 0;
        }
        BitReader br{data, end};
        int eobrun = 0, until_restart = restart;
        auto maybe_restart = [&]() {
            if (restart && until_restart == 0) { skip_to_restart(br); for (int k = 0; k < ns; k++) sc[k]->pred = 0; eobrun = 0; until_restart = restart; }
            if (restart) until_restart--;
        };
        if (ns == 1 && (progressive || comps.size() == 1)) { // non-interleaved: the component's own block raster
            Comp &c = *sc[0];
            for (int by = 0; by < c.cbh; by++) for (int bx = 0; bx < c.cbw; bx++) {
                maybe_restart();
                if (!block(br, c, c.coef.data() + ((size_t)by * c.bw + bx) * 64, ss, se, ah, al, eobrun)) return false;
            }
        } else {
            for (int my = 0; my < mcuy; my++) for (int mx = 0; mx < mcux; mx++) {
                maybe_restart();
                for (int k = 0; k < ns; k++) { Comp &c = *sc[k];
                    for (int by = 0; by < c.v; by++) for (int bx = 0; bx < c.h; bx++)
                        if (!block(br, c, c.coef.data() + ((size_t)(my * c.v + by) * c.bw + (mx * c.h + bx)) * 64, ss, se, ah, al, eobrun)) return false; }
            }
        }
        return true;
    }
};

} // namespace

// -> RGB8 (3 components) or R8 (grey), row-major; false + err on anything this reader does not handle
bool decode_jpeg(const uint8_t *d, size_t n, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, int &channels, std::string &err) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { err = "not a JPEG stream"; return false; }
    Decoder D;
    bool have_scan = false;
    size_t p = 2;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) { p++; continue; } // entropy-coded bytes of the scan just decoded: walk on to the next marker
        uint8_t m = d[p + 1];
        if (m == 0xFF) { p++; continue; }
        if (m == 0x00 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; } // stuffed byte / restart marker inside scan data
        if (m == 0xD9) break;
        size_t len = ((size_t)d[p + 2] << 8) | d[p + 3];
        if (len < 2 || p + 2 + len > n) { err = "truncated JPEG segment"; return false; }
        const uint8_t *s = d + p + 4; size_t sl = len - 2;
        if (m == 0xDB) { // quantisation tables
            size_t i = 0;
            while (i < sl) { int pq = s[i] >> 4, tq = s[i] & 15; i++; if (tq > 3 || i + (pq ? 128 : 64) > sl) { err = "bad JPEG DQT"; return false; }
                for (int k = 0; k < 64; k++) { D.qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i]; i += pq ? 2 : 1; } D.have_qt[tq] = true; }
        } else if (m == 0xC4) { // Huffman tables
            size_t i = 0;
            while (i + 17 <= sl) { int tc = s[i] >> 4, th = s[i] & 15; int total = 0; for (int k = 0; k < 16; k++) total += s[i + 1 + k];
                if (th > 3 || tc > 1 || total > 256 || i + 17 + (size_t)total > sl) { err = "bad JPEG DHT"; return false; }
                (tc ? D.hac : D.hdc)[th].build(s + i + 1, s + i + 17); i += 17 + (size_t)total; }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { // baseline / extended sequential / progressive, Huffman
            if (sl < 6 || s[0] != 8) { err = "JPEG sample precision other than 8 bits"; return false; }
            D.progressive = m == 0xC2;
            D.H = (s[1] << 8) | s[2]; D.W = (s[3] << 8) | s[4]; int nc = s[5];
            if ((nc != 1 && nc != 3) || sl < (size_t)(6 + 3 * nc) || D.W == 0 || D.H == 0) { err = "JPEG with an unsupported component count"; return false; }
            if (D.W > 16384 || D.H > 16384 || (size_t)D.W * (size_t)D.H > ((size_t)1 << 26)) { err = "JPEG extent beyond 16384 / 64 Mpixel"; return false; } // a 100-byte header must not ask for gigabytes of coefficients
            D.comps.resize((size_t)nc);
            for (int k = 0; k < nc; k++) { Comp &c = D.comps[k]; c.id = s[6 + 3 * k]; c.h = s[7 + 3 * k] >> 4; c.v = s[7 + 3 * k] & 15; c.tq = s[8 + 3 * k];
                if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) { err = "JPEG sampling factors beyond 2x2"; return false; } }
            if (nc == 1) D.comps[0].h = D.comps[0].v = 1;
            for (auto &c : D.comps) { D.hmax = c.h > D.hmax ? c.h : D.hmax; D.vmax = c.v > D.vmax ? c.v : D.vmax; }
            D.mcux = (D.W + 8 * D.hmax - 1) / (8 * D.hmax); D.mcuy = (D.H + 8 * D.vmax - 1) / (8 * D.vmax);
            for (auto &c : D.comps) {
                c.bw = D.mcux * c.h; c.bh = D.mcuy * c.v; c.w = c.bw * 8; c.hgt = c.bh * 8;
                int cw = (D.W * c.h + D.hmax - 1) / D.hmax, ch = (D.H * c.v + D.vmax - 1) / D.vmax;
                c.cbw = (cw + 7) / 8; c.cbh = (ch + 7) / 8;
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
        } else if (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC) { err = "lossless / hierarchical / arithmetic-coded JPEG is not supported"; return false;
        } else if (m == 0xDD) { if (sl >= 2) D.restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (D.comps.empty()) { err = "JPEG scan before frame header"; return false; }
            if (!D.scan(s, sl, d + p + 2 + len, d + n)) { err = D.err; return false; }
            have_scan = true;
            if (!D.progressive) break; // baseline: the one scan is the image
        }
        p += 2 + len;
    }
    if (!have_scan) { err = "JPEG without a scan"; return false; }
    for (auto &c : D.comps) { // dequantise + inverse DCT
        c.plane.assign((size_t)c.w * c.hgt, 0);
        for (int by = 0; by < c.bh; by++) for (int bx = 0; bx < c.bw; bx++) {
            const int16_t *q = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
            float blk[64];
            for (int k = 0; k < 64; k++) blk[k] = (float)q[k] * D.qt[c.tq][k];
            idct8x8(blk, c.plane.data() + (size_t)(by * 8) * c.w + bx * 8, c.w);
        }
    }
    const int W = D.W, H = D.H, hmax = D.hmax, vmax = D.vmax;
    std::vector<Comp> &comps = D.comps;
    // assemble: upsample chroma (triangle filter), YCbCr -> RGB
    width = (uint32_t)W; height = (uint32_t)H; channels = (int)comps.size() == 1 ? 1 : 3;
    pixels.assign((size_t)W * H * channels, 0);
    if (comps.size() == 1) { for (int y = 0; y < H; y++) std::memcpy(pixels.data() + (size_t)y * W, comps[0].plane.data() + (size_t)y * comps[0].w, (size_t)W); return true; }
    auto sample = [&](const Comp &c, int x, int y) -> float { // component value at full-resolution pixel (x, y)
        if (c.h == hmax && c.v == vmax) return (float)c.plane[(size_t)y * c.w + x];
        // position in the component's own grid (pixel centres): triangle weights 3/4, 1/4 along each subsampled axis
        float fx = c.h == hmax ? (float)x : ((float)x + 0.5f) * 0.5f - 0.5f, fy = c.v == vmax ? (float)y : ((float)y + 0.5f) * 0.5f - 0.5f;
        int cw = (W * c.h + hmax - 1) / hmax, ch = (H * c.v + vmax - 1) / vmax;
        int x0 = (int)std::floor(fx), y0 = (int)std::floor(fy); float ax = fx - (float)x0, ay = fy - (float)y0;
        auto at = [&](int xx, int yy) { xx = xx < 0 ? 0 : (xx >= cw ? cw - 1 : xx); yy = yy < 0 ? 0 : (yy >= ch ? ch - 1 : yy); return (float)c.plane[(size_t)yy * c.w + xx]; };
        float top = at(x0, y0) * (1 - ax) + at(x0 + 1, y0) * ax, bot = at(x0, y0 + 1) * (1 - ax) + at(x0 + 1, y0 + 1) * ax;
        return top * (1 - ay) + bot * ay;
    };
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        float Y = sample(comps[0], x, y), Cb = sample(comps[1], x, y) - 128.0f, Cr = sample(comps[2], x, y) - 128.0f;
        float rgb[3] = {Y + 1.402f * Cr, Y - 0.344136f * Cb - 0.714136f * Cr, Y + 1.772f * Cb};
        for (int k = 0; k < 3; k++) { int q = (int)std::lrintf(rgb[k]); pixels[((size_t)y * W + x) * 3 + k] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q)); }
    }
    return true;
}

} // namespace art
