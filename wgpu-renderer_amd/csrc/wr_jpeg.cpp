// wr_jpeg.cpp -- JPEG decode for the GLB importer (host only).
//
// The reference decodes images with the `image` crate (resources.rs:28 -> gltf::import),
// whose JPEG path is jpeg-decoder 0.3.0 (Cargo.lock); that crate is not part of the reference
// checkout, so nothing pins these texels to the reference's: PARITY UNPINNED for JPEG input.
// The pipeline below is the integer one that family of decoders shares: Huffman decode of
// baseline, extended-sequential and progressive 8-bit streams, dequantisation, a 12-bit
// fixed-point 8x8 inverse DCT, "fancy" (triangle-filter) chroma upsampling for 2:1 ratios and
// 20-bit fixed-point BT.601 YCbCr -> RGB.  Tests compare it with libjpeg (through PIL) within a
// small tolerance.  Output is RGBA8 with alpha 255 (resources.rs:184-191 pads RGB).
// One-channel and four-channel JPEGs are rejected: the reference asserts 3 or 4 channels
// (resources.rs:51) and `image` gives L8 / CMYK for those.

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wr_abi.h"

namespace {

const uint8_t kZigzag[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // padding: a corrupt run length can step past 63
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
    bool present = false;
    uint8_t values[256];
    int mincode[17], maxcode[18], valptr[17];  // per code length 1..16
    void build(const uint8_t counts[16], const uint8_t* vals, int nvals) {
        memcpy(values, vals, (size_t)nvals);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;          // Huffman table selectors of the current scan
    int bw = 0, bh = 0;          // blocks allocated (padded to whole MCUs)
    int w = 0, hgt = 0;          // samples
    int dc_pred = 0;
    std::vector<int16_t> coef;   // bw * bh * 64, natural (row-major) order within a block
    std::vector<uint8_t> pix;    // bw*8 x bh*8 after the inverse transform
};

struct Decoder {
    const uint8_t* p;
    const uint8_t* end;
    std::string* err;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comp;
    int width = 0, height = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;
    int adobe_transform = -1;
    // bit reader
    uint32_t bits = 0;
    int nbits = 0;
    bool hit_marker = false;
    int eobrun = 0;

    int fail(const char* m, int code = WR_ERR_INVALID_ARGUMENT) { *err = std::string("jpeg: ") + m; return code; }

    void reset_bits() { bits = 0; nbits = 0; hit_marker = false; }
    void fill() {
        while (nbits <= 24) {
            uint32_t b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    uint8_t n = p + 1 < end ? p[1] : 0xD9;
                    if (n == 0) { p += 2; }
                    else { hit_marker = true; b = 0; }  // a marker: feed zeros, leave p on it
                } else {
                    ++p;
                }
            }
            bits |= b << (24 - nbits);
            nbits += 8;
        }
    }
    int get_bit() { if (nbits < 1) fill(); int b = (int)(bits >> 31); bits <<= 1; --nbits; return b; }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        int v = (int)(bits >> (32 - n));
        bits <<= n; nbits -= n;
        return v;
    }
    static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
    int decode(const Huff& h) {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | get_bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
                return h.values[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }

    // ------------------------------------------------------------ block decoders
    bool block_baseline(Component& c, int16_t* blk) {
        const Huff& hd = dc[c.td]; const Huff& ha = ac[c.ta];
        int t = decode(hd);
        if (t < 0 || t > 15) return false;
        int diff = t ? extend(get_bits(t), t) : 0;
        c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)diff);  // wraps on corrupt data
        blk[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            int rs = decode(ha);
            if (rs < 0) return false;
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
            } else {
                k += r;
                blk[kZigzag[k]] = (int16_t)extend(get_bits(s), s);
                ++k;
            }
        }
        return true;
    }
    bool block_dc_first(Component& c, int16_t* blk, int al) {
        int t = decode(dc[c.td]);
        if (t < 0 || t > 15) return false;
        int diff = t ? extend(get_bits(t), t) : 0;
        c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)diff);
        blk[0] = (int16_t)((long long)c.dc_pred * (1 << al));
        return true;
    }
    void block_dc_refine(int16_t* blk, int al) { if (get_bit()) blk[0] = (int16_t)(blk[0] | (1 << al)); }
    bool block_ac_first(Component& c, int16_t* blk, int ss, int se, int al) {
        if (eobrun > 0) { --eobrun; return true; }
        const Huff& ha = ac[c.ta];
        for (int k = ss; k <= se;) {
            int rs = decode(ha);
            if (rs < 0) return false;
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += get_bits(r);
                    break;
                }
                k += 16;
            } else {
                k += r;
                blk[kZigzag[k]] = (int16_t)(extend(get_bits(s), s) * (1 << al));
                ++k;
            }
        }
        return true;
    }
    bool block_ac_refine(Component& c, int16_t* blk, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        const Huff& ha = ac[c.ta];
        int k = ss;
        if (eobrun <= 0) {
            for (; k <= se;) {
                int rs = decode(ha);
                if (rs < 0) return false;
                int r = rs >> 4, s = rs & 15, val = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1 << r);
                        if (r) eobrun += get_bits(r);
                        break;  // the rest of the block is refined below as part of the run
                    }
                    // r == 15: skip 16 zero coefficients (refining the non-zero ones passed)
                } else {
                    if (s != 1) return false;
                    val = get_bit() ? p1 : m1;
                }
                while (k <= se) {
                    int16_t* co = &blk[kZigzag[k++]];
                    if (*co != 0) {
                        if (get_bit() && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
                    } else {
                        if (r == 0) { if (val) *co = (int16_t)val; break; }
                        --r;
                    }
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t* co = &blk[kZigzag[k]];
                if (*co != 0 && get_bit() && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
            }
            --eobrun;
        }
        return true;
    }

    // ------------------------------------------------------------ one scan
    int scan(const std::vector<int>& sc, int ss, int se, int ah, int al) {
        reset_bits();
        eobrun = 0;
        for (Component& c : comp) c.dc_pred = 0;
        const bool interleaved = sc.size() > 1;
        int units_x, units_y;
        if (interleaved) { units_x = mcus_x; units_y = mcus_y; }
        else {
            const Component& c = comp[(size_t)sc[0]];
            units_x = (c.w + 7) / 8; units_y = (c.hgt + 7) / 8;
        }
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        for (int uy = 0; uy < units_y; ++uy)
            for (int ux = 0; ux < units_x; ++ux) {
                for (int ci : sc) {
                    Component& c = comp[(size_t)ci];
                    const int nh = interleaved ? c.h : 1, nv = interleaved ? c.v : 1;
                    for (int by = 0; by < nv; ++by)
                        for (int bx = 0; bx < nh; ++bx) {
                            const int gx = ux * nh + bx, gy = uy * nv + by;
                            if (gx >= c.bw || gy >= c.bh) return fail("block outside the image");
                            int16_t* blk = &c.coef[((size_t)gy * c.bw + gx) * 64];
                            bool ok;
                            if (!progressive) ok = block_baseline(c, blk);
                            else if (ss == 0) { if (ah == 0) ok = block_dc_first(c, blk, al); else { block_dc_refine(blk, al); ok = true; } }
                            else ok = ah == 0 ? block_ac_first(c, blk, ss, se, al) : block_ac_refine(c, blk, ss, se, al);
                            if (!ok) return fail("bad Huffman code");
                        }
                }
                if (--todo <= 0) {
                    // restart: byte-align, expect RSTn, reset predictors
                    reset_bits();
                    while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) {
                        if (p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF) break;  // another marker: the scan ends
                        ++p;
                    }
                    if (p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) p += 2;
                    for (Component& c : comp) c.dc_pred = 0;
                    eobrun = 0;
                    todo = restart_interval;
                }
            }
        // leave p at the next marker
        while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) ++p;
        return WR_OK;
    }

    // ------------------------------------------------------------ inverse DCT (12-bit fixed point)
    // 64-bit intermediates: the same values as 32-bit ones for any stream an encoder can write,
    // and no overflow for corrupt coefficients.
    typedef long long i64;
    static inline i64 f2f(double x) { return (i64)(x * 4096 + 0.5); }
    static inline uint8_t clamp8(i64 x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    static void idct_1d(i64 s0, i64 s1, i64 s2, i64 s3, i64 s4, i64 s5, i64 s6, i64 s7, i64& x0, i64& x1, i64& x2, i64& x3,
                        i64& t0, i64& t1, i64& t2, i64& t3) {
        i64 p2 = s2, p3 = s6;
        i64 p1 = (p2 + p3) * f2f(0.5411961);
        t2 = p1 + p3 * f2f(-1.847759065);
        t3 = p1 + p2 * f2f(0.765366865);
        p2 = s0; p3 = s4;
        t0 = (p2 + p3) * 4096;
        t1 = (p2 - p3) * 4096;
        x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = t0 + t2; i64 p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
        i64 p5 = (p3 + p4) * f2f(1.175875602);
        t0 = t0 * f2f(0.298631336);
        t1 = t1 * f2f(2.053119869);
        t2 = t2 * f2f(3.072711026);
        t3 = t3 * f2f(1.501321110);
        p1 = p5 + p1 * f2f(-0.899976223);
        p2 = p5 + p2 * f2f(-2.562915447);
        p3 = p3 * f2f(-1.961570560);
        p4 = p4 * f2f(-0.390180644);
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    }
    static void idct_block(uint8_t* out, int stride, const int16_t* coef, const uint16_t* q) {
        i64 d[64], val[64];
        for (int i = 0; i < 64; ++i) d[i] = (i64)coef[i] * (i64)q[i];
        for (int i = 0; i < 8; ++i) {
            const i64* c = d + i;
            i64* v = val + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                i64 dcterm = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
            } else {
                i64 x0, x1, x2, x3, t0, t1, t2, t3;
                idct_1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], x0, x1, x2, x3, t0, t1, t2, t3);
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
                v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
                v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
                v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
            }
        }
        for (int i = 0; i < 8; ++i) {
            const i64* v = val + i * 8;
            uint8_t* o = out + (size_t)i * stride;
            i64 x0, x1, x2, x3, t0, t1, t2, t3;
            idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
            const i64 bias = 65536 + (128 << 17);
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
    }

    // ------------------------------------------------------------ markers
    int parse() {
        if (end - p < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG");
        p += 2;
        for (;;) {
            while (p < end && *p != 0xFF) ++p;
            while (p < end && *p == 0xFF) ++p;
            if (p >= end) return fail("truncated");
            const int m = *p++;
            if (m == 0xD9) break;  // EOI
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (end - p < 2) return fail("truncated");
            const size_t len = ((size_t)p[0] << 8) | p[1];
            if (len < 2 || (size_t)(end - p) < len) return fail("truncated segment");
            const uint8_t* s = p + 2;
            const uint8_t* se_ = p + len;
            p += len;
            if (m == 0xDB) {  // DQT
                while (s < se_) {
                    const int pq = s[0] >> 4, tq = s[0] & 15;
                    ++s;
                    if (tq > 3 || pq > 1 || se_ - s < 64 * (pq + 1)) return fail("bad quantisation table");
                    for (int i = 0; i < 64; ++i) {
                        qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[0] << 8) | s[1]) : s[0];
                        s += pq + 1;
                    }
                    qt_present[tq] = true;
                }
            } else if (m == 0xC4) {  // DHT
                while (s < se_) {
                    if (se_ - s < 17) return fail("bad Huffman table");
                    const int tc = s[0] >> 4, th = s[0] & 15;
                    int n = 0;
                    for (int i = 0; i < 16; ++i) n += s[1 + i];
                    if (tc > 1 || th > 3 || n > 256 || se_ - s < 17 + n) return fail("bad Huffman table");
                    (tc ? ac[th] : dc[th]).build(s + 1, s + 17, n);
                    s += 17 + n;
                }
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {  // SOF0/1/2
                if (have_frame) return fail("second frame header");
                if (se_ - s < 6) return fail("bad frame header");
                if (s[0] != 8) return fail("only 8-bit samples are decoded", WR_ERR_UNSUPPORTED);
                progressive = m == 0xC2;
                height = (s[1] << 8) | s[2]; width = (s[3] << 8) | s[4];
                const int nc = s[5];
                if (width <= 0 || height <= 0) return fail("empty image");
                if (width > 16384 || height > 16384) return fail("image larger than 16384 pixels", WR_ERR_UNSUPPORTED);
                if (nc == 1) return fail("one-channel image: the reference takes 3 or 4 channels (resources.rs:51)", WR_ERR_UNSUPPORTED);
                if (nc != 3) return fail("four-channel (CMYK) JPEG", WR_ERR_UNSUPPORTED);
                if (se_ - s < 6 + 3 * nc) return fail("bad frame header");
                comp.resize((size_t)nc);
                for (int i = 0; i < nc; ++i) {
                    Component& c = comp[(size_t)i];
                    c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
                    if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad sampling factors");
                    if (c.h > hmax) hmax = c.h;
                    if (c.v > vmax) vmax = c.v;
                }
                mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
                mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
                for (Component& c : comp) {
                    c.w = (width * c.h + hmax - 1) / hmax;
                    c.hgt = (height * c.v + vmax - 1) / vmax;
                    c.bw = mcus_x * c.h; c.bh = mcus_y * c.v;
                    c.coef.assign((size_t)c.bw * c.bh * 64, 0);
                }
                have_frame = true;
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                return fail("lossless, hierarchical or arithmetic-coded JPEG", WR_ERR_UNSUPPORTED);
            } else if (m == 0xDD) {  // DRI
                if (se_ - s < 2) return fail("bad restart interval");
                restart_interval = (s[0] << 8) | s[1];
            } else if (m == 0xEE) {  // APP14 "Adobe"
                if (se_ - s >= 12 && !memcmp(s, "Adobe", 5)) adobe_transform = s[11];
            } else if (m == 0xDA) {  // SOS
                if (!have_frame) return fail("scan before the frame header");
                if (se_ - s < 1) return fail("bad scan header");
                const int ns = s[0];
                if (ns < 1 || ns > (int)comp.size() || se_ - s < 1 + 2 * ns + 3) return fail("bad scan header");
                std::vector<int> sc;
                for (int i = 0; i < ns; ++i) {
                    int ci = -1;
                    for (size_t k = 0; k < comp.size(); ++k) if (comp[k].id == s[1 + 2 * i]) ci = (int)k;
                    if (ci < 0) return fail("scan names an unknown component");
                    comp[(size_t)ci].td = s[2 + 2 * i] >> 4; comp[(size_t)ci].ta = s[2 + 2 * i] & 15;
                    if (comp[(size_t)ci].td > 3 || comp[(size_t)ci].ta > 3) return fail("bad table selector");
                    sc.push_back(ci);
                }
                int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
                if (!progressive) { ss = 0; se = 63; ah = 0; al = 0; }
                if (ss > se || se > 63 || al > 13 || (ss == 0 && se != 0 && progressive) || (ss > 0 && ns != 1)) return fail("bad spectral selection");
                for (int ci : sc) {
                    const Component& c = comp[(size_t)ci];
                    const bool need_dc = !progressive || (ss == 0 && ah == 0);
                    const bool need_ac = !progressive || ss > 0;
                    if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present)) return fail("scan uses a Huffman table that was not defined");
                }
                int rc = scan(sc, ss, se, ah, al);
                if (rc != WR_OK) return rc;
            }
            // other segments (APPn, COM) are skipped
        }
        if (!have_frame) return fail("no frame header");
        return WR_OK;
    }

    // ------------------------------------------------------------ samples -> RGBA
    // Row r of component c at full width, upsampled horizontally (and vertically between the
    // two nearest component rows) with the triangle filter where the ratio is 2, by
    // replication otherwise.
    void component_row(const Component& c, int row, std::vector<uint8_t>& out, std::vector<uint8_t>& tmp) const {
        const int hs = hmax / c.h, vs = vmax / c.v;
        const int stride = c.bw * 8;
        const int cw = c.w;
        const uint8_t* near_;
        const uint8_t* far_;
        if (vs == 2) {
            // sample row `row` lies between component rows: the nearer one weighs 3, the other 1
            const int cy = row >> 1;
            const int other = (row & 1) ? cy + 1 : cy - 1;
            const int oy = other < 0 ? 0 : (other >= c.hgt ? c.hgt - 1 : other);
            near_ = &c.pix[(size_t)cy * stride];
            far_ = &c.pix[(size_t)oy * stride];
        } else {
            const int cy = row / vs;
            near_ = far_ = &c.pix[(size_t)(cy >= c.hgt ? c.hgt - 1 : cy) * stride];
        }
        out.assign((size_t)cw * hs + 2, 0);
        if (hs == 2 && vs == 2) {
            if (cw == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return; }
            int t1 = 3 * near_[0] + far_[0];
            out[0] = (uint8_t)((t1 + 2) >> 2);
            for (int i = 1; i < cw; ++i) {
                const int t0 = t1;
                t1 = 3 * near_[i] + far_[i];
                out[(size_t)i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                out[(size_t)i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
            }
            out[(size_t)cw * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
            return;
        }
        const uint8_t* src = near_;
        if (vs == 2) {
            tmp.resize((size_t)cw);
            for (int i = 0; i < cw; ++i) tmp[(size_t)i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
            src = tmp.data();
        }
        if (hs == 1) { memcpy(out.data(), src, (size_t)cw); return; }
        if (hs == 2) {
            if (cw == 1) { out[0] = out[1] = src[0]; return; }
            out[0] = src[0];
            out[1] = (uint8_t)((src[0] * 3 + src[1] + 2) >> 2);
            int i;
            for (i = 1; i < cw - 1; ++i) {
                const int n = 3 * src[i] + 2;
                out[(size_t)i * 2] = (uint8_t)((n + src[i - 1]) >> 2);
                out[(size_t)i * 2 + 1] = (uint8_t)((n + src[i + 1]) >> 2);
            }
            out[(size_t)i * 2] = (uint8_t)((src[cw - 1] * 3 + src[cw - 2] + 2) >> 2);
            out[(size_t)i * 2 + 1] = src[cw - 1];
            return;
        }
        for (int i = 0; i < cw; ++i)
            for (int k = 0; k < hs; ++k) out[(size_t)i * hs + k] = src[i];
    }

    int finish(uint32_t* ow, uint32_t* oh, std::vector<uint8_t>& rgba) {
        for (Component& c : comp) {
            if (!qt_present[c.tq]) return fail("component uses a quantisation table that was not defined");
            if (hmax % c.h || vmax % c.v) return fail("fractional sampling ratio", WR_ERR_UNSUPPORTED);
            c.pix.assign((size_t)c.bw * 8 * c.bh * 8, 0);
            for (int by = 0; by < c.bh; ++by)
                for (int bx = 0; bx < c.bw; ++bx)
                    idct_block(&c.pix[((size_t)by * 8) * (c.bw * 8) + (size_t)bx * 8], c.bw * 8, &c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq]);
        }
        // Adobe transform 0 with three components means the samples are already RGB
        const bool ycc = adobe_transform != 0;
        rgba.assign((size_t)width * height * 4, 255);
        std::vector<uint8_t> row[3], tmp;
        for (int y = 0; y < height; ++y) {
            for (int k = 0; k < 3; ++k) component_row(comp[(size_t)k], y, row[k], tmp);
            uint8_t* o = &rgba[(size_t)y * width * 4];
            for (int x = 0; x < width; ++x) {
                int a = row[0][(size_t)x], b = row[1][(size_t)x], c = row[2][(size_t)x];
                if (ycc) {
                    const int yf = (a << 20) + (1 << 19), cb = b - 128, cr = c - 128;
                    int r = yf + cr * (((int)(1.40200f * 4096.0f + 0.5f)) << 8);
                    int g = yf + cr * -(((int)(0.71414f * 4096.0f + 0.5f)) << 8) + ((cb * -(((int)(0.34414f * 4096.0f + 0.5f)) << 8)) & (int)0xffff0000);
                    int bl = yf + cb * (((int)(1.77200f * 4096.0f + 0.5f)) << 8);
                    a = clamp8(r >> 20); b = clamp8(g >> 20); c = clamp8(bl >> 20);
                }
                o[4 * x] = (uint8_t)a; o[4 * x + 1] = (uint8_t)b; o[4 * x + 2] = (uint8_t)c;
            }
        }
        *ow = (uint32_t)width; *oh = (uint32_t)height;
        return WR_OK;
    }
};

}  // namespace

// Returns WR_OK, WR_ERR_UNSUPPORTED (a layout the reference rejects or this decoder does not
// handle) or WR_ERR_INVALID_ARGUMENT (corrupt data).
int wr_decode_jpeg(const uint8_t* data, size_t n, uint32_t* w, uint32_t* h, std::vector<uint8_t>& rgba, std::string& err) {
    Decoder d;
    memset(d.qt, 0, sizeof(d.qt));
    d.p = data; d.end = data + n; d.err = &err;
    int rc = d.parse();
    if (rc != WR_OK) return rc;
    return d.finish(w, h, rgba);
}
