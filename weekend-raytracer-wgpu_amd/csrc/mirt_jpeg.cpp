// mirt_jpeg.cpp — the JPEG -> RGB8 step of `Texture::new_from_image` (reference src/raytracer/texture.rs:21-46), host side.
//
// The reference decodes `assets/earthmap.jpeg` (baseline, 4:4:4) and `assets/moon.jpeg` (progressive, 4:4:4) with the `image`
// crate (0.24, jpeg-decoder) and turns every pixel into `inv_255 * (p as f32)`.  This file is an own decoder for that step, so
// that a host needs nothing but libmirt.so to get from the reference's asset files to the texel table of MirtScene:
//   * baseline and progressive DCT, 8-bit, Huffman (SOF0 / SOF1 / SOF2), interleaved and non-interleaved scans, restart
//     intervals, 1 (grey) or 3 (YCbCr / RGB by the Adobe marker) components;
//   * chroma sampling 4:4:4, 4:2:2 and 4:2:0 with the triangle ("fancy") upsampling filter, other integral ratios by
//     replication; arithmetic coding, 12-bit, CMYK, lossless and hierarchical files are refused.
// The arithmetic behind the entropy decoder is the de-facto standard one — the 13-bit "slow integer" inverse DCT, the
// 16-bit fixed-point YCbCr -> RGB tables and the triangle upsampling of the Independent JPEG Group's published design, which
// libjpeg-turbo (the decoder behind Pillow) also implements — so that the output can be PINNED: tests/test_jpeg.py
// requires bit-identical RGB8 against Pillow on generated files of every supported kind and against the committed decodes
// of the reference's two assets.  Against the `image` crate itself the result stays DECODER-UNPINNED (its IDCT may differ
// by one unit in some texels; the reference holds nothing that pins either): DESIGN.md §2.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mirt.h"

namespace {

enum : int { kOk = 0 };

const uint8_t kZigzag[64] = { 0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5, 12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13,  6,  7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Huff {
    bool     present = false;
    uint8_t  bits[17] = {};
    uint8_t  vals[256] = {};
    int32_t  maxcode[18];
    int32_t  valptr[17];
    uint16_t look[512];          // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
    bool build()                 // false: the code lengths do not describe a prefix code
    {
        uint16_t codes[256];
        uint8_t sizes[256];
        int n = 0;
        for (int l = 1; l <= 16; ++l)
            for (int i = 0; i < bits[l]; ++i) sizes[n++] = (uint8_t)l;
        uint32_t code = 0;
        int k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k - (int)code;               // symbol index = code + valptr[l]
            while (k < n && sizes[k] == l) codes[k++] = (uint16_t)code++;
            if (code > (1u << l)) return false;
            maxcode[l] = bits[l] ? (int32_t)code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        std::memset(look, 0, sizeof look);
        for (int i = 0; i < n; ++i) {
            if (sizes[i] > 9) continue;
            const int shift = 9 - sizes[i];
            for (int f = 0; f < (1 << shift); ++f) look[(codes[i] << shift) | f] = (uint16_t)((sizes[i] << 8) | vals[i]);
        }
        present = true;
        return true;
    }
};

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int      n = 0;
    bool     at_marker = false;   // a marker stops the entropy-coded segment: zeros are supplied from there on
    void fill()
    {
        while (n <= 48) {
            uint32_t b = 0;
            if (!at_marker && p < end) {
                b = *p;
                if (b == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;            // stuffed byte
                    else { at_marker = true; b = 0; }                   // leave p at the marker
                } else {
                    ++p;
                }
            } else {
                at_marker = true;
            }
            acc = (acc << 8) | b;
            n += 8;
        }
    }
    uint32_t peek(int k) { if (n < k) fill(); return (uint32_t)(acc >> (n - k)) & ((1u << k) - 1u); }
    void     skip(int k) { n -= k; }
    uint32_t get(int k) { if (k <= 0) return 0; if (k > 16) k = 16; const uint32_t v = peek(k); skip(k); return v; }   // > 16 only in corrupt files
    void     reset() { acc = 0; n = 0; at_marker = false; }
};

inline int extend(uint32_t v, int s) { if (s > 16) s = 16; return (v < (1u << (s - 1))) ? (int)v - (1 << s) + 1 : (int)v; }

inline int decode_symbol(Bits& b, const Huff& h)
{
    const uint32_t la = b.peek(9);
    const uint16_t e = h.look[la];
    if (e) { b.skip(e >> 8); return e & 0xff; }
    int32_t code = (int32_t)b.peek(16);
    for (int l = 10; l <= 16; ++l) {
        const int32_t c = code >> (16 - l);
        if (c <= h.maxcode[l]) { b.skip(l); return h.vals[(c + h.valptr[l]) & 0xff]; }
    }
    b.skip(16);
    return 0;       // corrupt data: the IJG decoder also carries on with a zero
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;     // allocated (whole MCUs)
    int real_w = 0, real_h = 0;         // samples that belong to the image: ceil(width * h / hmax), ceil(height * v / vmax)
    int pred = 0;
    std::vector<int16_t> coef;          // [blocks_h][blocks_w][64], natural order
    std::vector<uint8_t> plane;         // [blocks_h * 8][blocks_w * 8] after the inverse DCT
};

struct Decoder {
    const uint8_t* data;
    size_t len;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false;
    bool adobe = false;
    int  adobe_transform = 0;
    bool jfif = false;
    uint16_t quant[4][64] = {};
    bool have_quant[4] = {};
    Huff dc[4], ac[4];
    Component comp[3];
    int restart_interval = 0;
    int eobrun = 0;
    const char* err = nullptr;

    int fail(const char* m) { err = m; return MIRT_ERR_IMAGE_DECODE; }

    static uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

    // ---- markers ----
    int parse(bool header_only)
    {
        if (len < 4 || data[0] != 0xff || data[1] != 0xd8) return fail("not a JPEG file (no SOI)");
        size_t pos = 2;
        for (;;) {
            while (pos < len && data[pos] != 0xff) ++pos;         // tolerate garbage between segments
            while (pos < len && data[pos] == 0xff) ++pos;         // fill bytes
            if (pos >= len) return fail("unexpected end of file");
            const uint8_t mk = data[pos++];
            if (mk == 0xd9) break;                                // EOI
            if (mk == 0x01 || (mk >= 0xd0 && mk <= 0xd7)) continue;
            if (pos + 2 > len) return fail("truncated segment");
            const size_t seg = be16(data + pos);
            if (seg < 2 || pos + seg > len) return fail("bad segment length");
            const uint8_t* s = data + pos + 2;
            const size_t n = seg - 2;
            switch (mk) {
            case 0xc0: case 0xc1: case 0xc2: {
                if (have_frame) return fail("more than one frame");
                if (n < 6) return fail("bad SOF");
                if (s[0] != 8) return fail("only 8-bit samples are supported");
                height = (int)be16(s + 1); width = (int)be16(s + 3); ncomp = s[5];
                if (width <= 0 || height <= 0) return fail("empty image");
                if (ncomp != 1 && ncomp != 3) return fail("only 1- and 3-component files are supported");
                if (n < 6 + 3 * (size_t)ncomp) return fail("bad SOF");
                for (int i = 0; i < ncomp; ++i) {
                    comp[i].id = s[6 + 3 * i];
                    comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i] & 3;
                    if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) return fail("bad sampling factor");
                    hmax = comp[i].h > hmax ? comp[i].h : hmax; vmax = comp[i].v > vmax ? comp[i].v : vmax;
                }
                progressive = mk == 0xc2;
                have_frame = true;
                if (header_only) return kOk;
                // untrusted input: the coefficient and pixel planes are allocated from the header alone, before any entropy data is
                // read.  Bound the frame by 2^28 pixels AND by the file: the densest legal coding (two 1-bit codes per block, 32x32-pixel
                // MCUs of 18 blocks) is about 230 pixels per byte, so 4096 pixels per byte of file refuses only headers that lie
                // (a 200-byte file claiming 16384 x 16384 would otherwise allocate 1.5 GB).
                if ((uint64_t)width * height > (1ull << 28) || (uint64_t)width * height > 4096ull * (uint64_t)len + 65536ull) return fail("image too large");
                const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
                for (int i = 0; i < ncomp; ++i) {
                    Component& c = comp[i];
                    c.blocks_w = mcux * c.h; c.blocks_h = mcuy * c.v;
                    c.real_w = (width * c.h + hmax - 1) / hmax; c.real_h = (height * c.v + vmax - 1) / vmax;
                    c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
                }
                break;
            }
            case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                return fail("unsupported JPEG process (lossless, hierarchical or arithmetic coding)");
            case 0xc4: {                                          // DHT
                size_t o = 0;
                while (o < n) {
                    if (o + 17 > n) return fail("bad DHT");
                    const int tc = s[o] >> 4, th = s[o] & 15;
                    if (tc > 1 || th > 3) return fail("bad DHT");
                    Huff& h = tc ? ac[th] : dc[th];
                    int total = 0;
                    h.bits[0] = 0;
                    for (int l = 1; l <= 16; ++l) { h.bits[l] = s[o + l]; total += h.bits[l]; }
                    if (total > 256 || o + 17 + (size_t)total > n) return fail("bad DHT");
                    std::memset(h.vals, 0, sizeof h.vals);
                    std::memcpy(h.vals, s + o + 17, (size_t)total);
                    if (!h.build()) return fail("bad Huffman table");
                    o += 17 + (size_t)total;
                }
                break;
            }
            case 0xdb: {                                          // DQT
                size_t o = 0;
                while (o < n) {
                    const int pq = s[o] >> 4, tq = s[o] & 15;
                    if (tq > 3 || pq > 1) return fail("bad DQT");
                    const size_t need = pq ? 128 : 64;
                    if (o + 1 + need > n) return fail("bad DQT");
                    for (int k = 0; k < 64; ++k)
                        quant[tq][kZigzag[k]] = pq ? (uint16_t)be16(s + o + 1 + 2 * k) : s[o + 1 + k];
                    have_quant[tq] = true;
                    o += 1 + need;
                }
                break;
            }
            case 0xdd:
                if (n < 2) return fail("bad DRI");
                restart_interval = (int)be16(s);
                break;
            case 0xe0:
                if (n >= 5 && std::memcmp(s, "JFIF\0", 5) == 0) jfif = true;
                break;
            case 0xee:
                if (n >= 12 && std::memcmp(s, "Adobe", 5) == 0) { adobe = true; adobe_transform = s[11]; }
                break;
            case 0xda: {                                          // SOS
                if (!have_frame) return fail("scan before frame");
                if (header_only) return kOk;
                size_t used = 0;
                const int rc = scan(s, n, data + pos + seg, &used);
                if (rc != kOk) return rc;
                pos += seg + used;
                continue;
            }
            default:
                break;
            }
            pos += seg;
        }
        if (!have_frame) return fail("no frame");
        return kOk;
    }

    // ---- one scan ----
    int scan(const uint8_t* s, size_t n, const uint8_t* ecs, size_t* used)
    {
        if (n < 1) return fail("bad SOS");
        const int ns = s[0];
        if (ns < 1 || ns > ncomp || n < 1 + 2 * (size_t)ns + 3) return fail("bad SOS");
        Component* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = s[1 + 2 * i];
            sc[i] = nullptr;
            for (int c = 0; c < ncomp; ++c) if (comp[c].id == id) sc[i] = &comp[c];
            if (!sc[i]) return fail("scan names an unknown component");
            sc[i]->td = s[2 + 2 * i] >> 4; sc[i]->ta = s[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table selector");
        }
        int ss = s[1 + 2 * ns], se = s[2 + 2 * ns];
        const int ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
        if (!progressive) { ss = 0; se = 63; }
        if (ss > se || se > 63 || (progressive && ss == 0 && se != 0) || (progressive && ss > 0 && ns != 1) || al > 13) return fail("bad progression parameters");
        for (int i = 0; i < ns; ++i) {
            if ((ss == 0 && ah == 0) && !dc[sc[i]->td].present) return fail("missing DC Huffman table");
            if (se > 0 && !ac[sc[i]->ta].present) return fail("missing AC Huffman table");
        }

        Bits b;
        b.p = ecs; b.end = data + len;
        for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
        eobrun = 0;
        int mcus_x, mcus_y;
        if (ns == 1) { mcus_x = (sc[0]->real_w + 7) / 8; mcus_y = (sc[0]->real_h + 7) / 8; }
        else { mcus_x = (width + 8 * hmax - 1) / (8 * hmax); mcus_y = (height + 8 * vmax - 1) / (8 * vmax); }
        int until_restart = restart_interval;
        int expected_rst = 0;
        for (int my = 0; my < mcus_y; ++my) {
            for (int mx = 0; mx < mcus_x; ++mx) {
                if (restart_interval && until_restart == 0) {
                    // byte-align, find RSTn, reset the predictors
                    b.reset();
                    const uint8_t* q = b.p;
                    while (q + 1 < b.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) {
                        if (q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff) break;      // some other marker: give up resynchronising
                        ++q;
                    }
                    if (q + 1 < b.end && q[0] == 0xff && q[1] == 0xd0 + expected_rst) b.p = q + 2; else b.p = q, b.at_marker = true;
                    expected_rst = (expected_rst + 1) & 7;
                    for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                    eobrun = 0;
                    until_restart = restart_interval;
                }
                if (ns == 1) {
                    block(b, *sc[0], my, mx, ss, se, ah, al);
                } else {
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h) block(b, *sc[i], my * sc[i]->v + v, mx * sc[i]->h + h, ss, se, ah, al);
                }
                if (restart_interval) --until_restart;
            }
        }
        // the segment ends at the next marker that is not a restart marker (re-scanned from the start of the segment:
        // cheap, and independent of how far the bit reader pre-fetched)
        const uint8_t* q = ecs;
        while (q + 1 < data + len) {
            if (q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7)) break;
            ++q;
        }
        *used = (size_t)(q - ecs);
        return kOk;
    }

    void block(Bits& b, Component& c, int by, int bx, int ss, int se, int ah, int al)
    {
        int16_t* blk = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
        if (!progressive) {
            const int t = decode_symbol(b, dc[c.td]);
            const int diff = t ? extend(b.get(t), t) : 0;
            c.pred = (int)((unsigned)c.pred + (unsigned)diff);       // wraps instead of overflowing on a crafted stream (valid files: |DC| <= 2^15)
            blk[0] = (int16_t)c.pred;
            for (int k = 1; k < 64;) {
                const int rs = decode_symbol(b, ac[c.ta]);
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                k += r;
                if (k > 63) break;
                blk[kZigzag[k]] = (int16_t)extend(b.get(s), s);
                ++k;
            }
            return;
        }
        if (ss == 0) {
            if (ah == 0) {                                     // DC, first pass
                const int t = decode_symbol(b, dc[c.td]);
                const int diff = t ? extend(b.get(t), t) : 0;
                c.pred = (int)((unsigned)c.pred + (unsigned)diff);
                blk[0] = (int16_t)((unsigned)c.pred << al);
            } else if (b.get(1)) {                             // DC, refinement
                blk[0] = (int16_t)(blk[0] | (1 << al));
            }
            return;
        }
        if (ah == 0) {                                         // AC, first pass
            if (eobrun > 0) { --eobrun; return; }
            for (int k = ss; k <= se;) {
                const int rs = decode_symbol(b, ac[c.ta]);
                const int r = rs >> 4, s = rs & 15;
                if (s) {
                    k += r;
                    if (k > 63) break;
                    blk[kZigzag[k]] = (int16_t)(extend(b.get(s), s) * (1 << al));
                    ++k;
                } else if (r == 15) {
                    k += 16;
                } else {
                    eobrun = 1 << r;
                    if (r) eobrun += (int)b.get(r);
                    --eobrun;
                    break;
                }
            }
            return;
        }
        // AC, refinement
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                const int rs = decode_symbol(b, ac[c.ta]);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    s = b.get(1) ? p1 : m1;                    // a newly non-zero coefficient (size is 1)
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += (int)b.get(r);
                    break;                                     // end of band: the rest of the block only gets correction bits
                }
                // skip r still-zero coefficients, appending correction bits to the non-zero ones passed on the way
                while (k <= se) {
                    int16_t& co = blk[kZigzag[k]];
                    if (co != 0) {
                        if (b.get(1) && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1));
                    } else {
                        if (--r < 0) break;
                    }
                    ++k;
                }
                if (s && k <= 63) blk[kZigzag[k]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t& co = blk[kZigzag[k]];
                if (co != 0 && b.get(1) && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1));
            }
            --eobrun;
        }
    }

    // ---- inverse DCT: 13-bit "slow integer" (Loeffler-Ligtenberg-Moschytz), two passes, the IJG arithmetic ----
    // (64-bit temporaries: identical results on valid files, and no signed overflow on corrupt coefficients)
    static inline int64_t descale(int64_t x, int n) { return (x + (1 << (n - 1))) >> n; }
    static inline uint8_t range_limit(int64_t x)
    {
        // the IJG post-IDCT table: x is taken modulo 1024 as a signed 10-bit value v, the sample is clamp(v + 128, 0, 255)
        int64_t v = x & 0x3ff;
        if (v >= 512) v -= 1024;
        v += 128;
        return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }

    static void idct_block(const int16_t* in, const uint16_t* q, uint8_t* out, int stride)
    {
        constexpr int CB = 13, P1 = 2;
        constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                          F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        int64_t ws[64];
        for (int c = 0; c < 8; ++c) {
            const int16_t* ip = in + c;
            const uint16_t* qp = q + c;
            if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
                const int64_t d = ((int64_t)ip[0] * qp[0]) * (1 << P1);
                for (int r = 0; r < 8; ++r) ws[8 * r + c] = d;
                continue;
            }
            int64_t z2 = (int64_t)ip[16] * qp[16], z3 = (int64_t)ip[48] * qp[48];
            int64_t z1 = (z2 + z3) * F0_541;
            int64_t tmp2 = z1 + z3 * (-F1_847);
            int64_t tmp3 = z1 + z2 * F0_765;
            z2 = (int64_t)ip[0] * qp[0]; z3 = (int64_t)ip[32] * qp[32];
            int64_t tmp0 = (z2 + z3) * (1 << CB);
            int64_t tmp1 = (z2 - z3) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = (int64_t)ip[56] * qp[56]; tmp1 = (int64_t)ip[40] * qp[40]; tmp2 = (int64_t)ip[24] * qp[24]; tmp3 = (int64_t)ip[8] * qp[8];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            ws[8 * 0 + c] = descale(tmp10 + tmp3, CB - P1); ws[8 * 7 + c] = descale(tmp10 - tmp3, CB - P1);
            ws[8 * 1 + c] = descale(tmp11 + tmp2, CB - P1); ws[8 * 6 + c] = descale(tmp11 - tmp2, CB - P1);
            ws[8 * 2 + c] = descale(tmp12 + tmp1, CB - P1); ws[8 * 5 + c] = descale(tmp12 - tmp1, CB - P1);
            ws[8 * 3 + c] = descale(tmp13 + tmp0, CB - P1); ws[8 * 4 + c] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; ++r) {
            const int64_t* w = ws + 8 * r;
            uint8_t* o = out + (size_t)r * stride;
            int64_t z2 = w[2], z3 = w[6];
            int64_t z1 = (z2 + z3) * F0_541;
            int64_t tmp2 = z1 + z3 * (-F1_847);
            int64_t tmp3 = z1 + z2 * F0_765;
            int64_t tmp0 = (w[0] + w[4]) * (1 << CB);
            int64_t tmp1 = (w[0] - w[4]) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            constexpr int S = CB + P1 + 3;
            o[0] = range_limit(descale(tmp10 + tmp3, S)); o[7] = range_limit(descale(tmp10 - tmp3, S));
            o[1] = range_limit(descale(tmp11 + tmp2, S)); o[6] = range_limit(descale(tmp11 - tmp2, S));
            o[2] = range_limit(descale(tmp12 + tmp1, S)); o[5] = range_limit(descale(tmp12 - tmp1, S));
            o[3] = range_limit(descale(tmp13 + tmp0, S)); o[4] = range_limit(descale(tmp13 - tmp0, S));
        }
    }

    int reconstruct()
    {
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            if (!have_quant[c.tq]) return fail("missing quantisation table");
            const int stride = c.blocks_w * 8;
            c.plane.assign((size_t)stride * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx)
                    idct_block(&c.coef[((size_t)by * c.blocks_w + bx) * 64], quant[c.tq], &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
        }
        return kOk;
    }

    // ---- chroma upsampling to full resolution (rows x width), the IJG triangle filter for 2:1 ----
    // row(y) of a component plane with the vertical neighbour rule of the filter: rows outside [0, real_h) repeat the edge row
    static const uint8_t* plane_row(const Component& c, int y)
    {
        y = y < 0 ? 0 : (y >= c.real_h ? c.real_h - 1 : y);
        return &c.plane[(size_t)y * c.blocks_w * 8];
    }

    void upsample_row(const Component& c, int y, std::vector<uint8_t>& out) const
    {
        const int hx = hmax / c.h, vx = vmax / c.v;
        out.resize((size_t)c.real_w * hx + 2);
        if (hx == 1 && vx == 1) {
            std::memcpy(out.data(), plane_row(c, y), (size_t)c.real_w);
        } else if (hx == 2 && vx == 1) {                                  // 4:2:2
            const uint8_t* in = plane_row(c, y);
            const int n = c.real_w;
            if (n == 1) { out[0] = out[1] = in[0]; return; }
            out[0] = in[0];
            out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            for (int i = 1; i < n - 1; ++i) {
                const int v = in[i] * 3;
                out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
                out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
            }
            out[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
            out[2 * n - 1] = in[n - 1];
        } else if (hx == 2 && vx == 2) {                                  // 4:2:0
            const int yi = y >> 1;
            const uint8_t* in0 = plane_row(c, yi);                        // the nearer input row
            const uint8_t* in1 = plane_row(c, (y & 1) ? yi + 1 : yi - 1); // the farther one
            const int n = c.real_w;
            if (n == 1) { const int t = in0[0] * 3 + in1[0]; out[0] = (uint8_t)((t * 4 + 8) >> 4); out[1] = (uint8_t)((t * 4 + 7) >> 4); return; }
            int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
            out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
            out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
            for (int i = 1; i < n - 1; ++i) {
                nextcol = in0[i + 1] * 3 + in1[i + 1];
                out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
            }
            out[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            out[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        } else {                                                          // other integral ratios: replication
            const uint8_t* in = plane_row(c, y / vx);
            for (int i = 0; i < c.real_w; ++i)
                for (int k = 0; k < hx; ++k) out[(size_t)i * hx + k] = in[i];
        }
    }

    int to_rgb(uint8_t* rgb)
    {
        for (int i = 0; i < ncomp; ++i)
            if (hmax % comp[i].h || vmax % comp[i].v) return fail("fractional sampling ratios are not supported");
        if (ncomp == 1) {
            for (int y = 0; y < height; ++y) {
                const uint8_t* in = plane_row(comp[0], y);
                uint8_t* o = rgb + (size_t)y * width * 3;
                for (int x = 0; x < width; ++x) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = in[x]; }
            }
            return kOk;
        }
        // colour space: Adobe transform 0 = RGB; otherwise YCbCr (JFIF, Adobe transform 1, or component ids that are not 'R','G','B')
        bool ycc = true;
        if (adobe) ycc = adobe_transform != 0;
        else if (!jfif && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = false;
        // the IJG fixed-point tables (16 fractional bits)
        int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
        auto fix = [](double x) { return (int32_t)(x * 65536.0 + 0.5); };
        for (int i = 0; i < 256; ++i) {
            const int32_t x = i - 128;
            cr_r[i] = (fix(1.40200) * x + 32768) >> 16;
            cb_b[i] = (fix(1.77200) * x + 32768) >> 16;
            cr_g[i] = -fix(0.71414) * x;
            cb_g[i] = -fix(0.34414) * x + 32768;
        }
        auto clamp8 = [](int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        std::vector<uint8_t> r0, r1, r2;
        for (int y = 0; y < height; ++y) {
            upsample_row(comp[0], y, r0); upsample_row(comp[1], y, r1); upsample_row(comp[2], y, r2);
            uint8_t* o = rgb + (size_t)y * width * 3;
            if (!ycc) {
                for (int x = 0; x < width; ++x) { o[3 * x] = r0[x]; o[3 * x + 1] = r1[x]; o[3 * x + 2] = r2[x]; }
                continue;
            }
            for (int x = 0; x < width; ++x) {
                const int32_t Y = r0[x], cb = r1[x], cr = r2[x];
                o[3 * x] = clamp8(Y + cr_r[cr]);
                o[3 * x + 1] = clamp8(Y + ((cb_g[cb] + cr_g[cr]) >> 16));
                o[3 * x + 2] = clamp8(Y + cb_b[cb]);
            }
        }
        return kOk;
    }
};

thread_local char g_jpeg_error[160] = "";

int report(const Decoder& d, int rc)
{
    if (rc != kOk) { std::strncpy(g_jpeg_error, d.err ? d.err : "JPEG decode failed", sizeof g_jpeg_error - 1); g_jpeg_error[sizeof g_jpeg_error - 1] = 0; }
    return rc;
}

}  // namespace

extern "C" {

const char* mirt_jpeg_last_error(void) { return g_jpeg_error; }

int mirt_jpeg_info(const uint8_t* data, size_t len, uint32_t* width, uint32_t* height)
{
    if (!data || !width || !height) { std::strncpy(g_jpeg_error, "null pointer", sizeof g_jpeg_error - 1); return MIRT_ERR_NULL_POINTER; }
    Decoder d;
    d.data = data; d.len = len;
    const int rc = d.parse(true);
    if (rc != kOk) return report(d, rc);
    *width = (uint32_t)d.width; *height = (uint32_t)d.height;
    return MIRT_OK;
}

int mirt_jpeg_decode_rgb8(const uint8_t* data, size_t len, uint8_t* rgb, size_t rgb_len)
{
    if (!data || !rgb) { std::strncpy(g_jpeg_error, "null pointer", sizeof g_jpeg_error - 1); return MIRT_ERR_NULL_POINTER; }
    Decoder d;
    d.data = data; d.len = len;
    int rc = d.parse(false);
    if (rc != kOk) return report(d, rc);
    if (rgb_len < (size_t)d.width * d.height * 3) { std::strncpy(g_jpeg_error, "output buffer too small", sizeof g_jpeg_error - 1); return MIRT_ERR_OUT_BUFFER; }
    rc = d.reconstruct();
    if (rc != kOk) return report(d, rc);
    return report(d, d.to_rgb(rgb));
}

/* texture.rs:30-41: `inv_255 * (p as f32)` per channel */
int mirt_rgb8_to_texels(const uint8_t* rgb, size_t n_pixels, float* texels)
{
    if (!rgb || !texels) return MIRT_ERR_NULL_POINTER;
    const float inv_255 = 1.0f / 255.0f;
    for (size_t i = 0; i < 3 * n_pixels; ++i) texels[i] = inv_255 * (float)rgb[i];
    return MIRT_OK;
}

}  // extern "C"
