/*
 * jpeg_decode.hpp — header-only baseline JPEG decoder and image loader for the host side.
 *
 * Mirrors the reference's `load_image_to_rgb(filename) -> Result<RgbImage, ImageError>`
 * (/root/reference/src/scene/image_loader.rs:5-8), which goes through `image 0.25.5` -> `zune-jpeg 0.4.14`.
 * That decoder is a third-party dependency and is not restated here; this one implements ITU-T T.81 directly:
 * sequential Huffman DCT (SOF0 / SOF1, 8-bit), 1 or 3 components, sampling factors 1..2 (chroma replicated),
 * restart intervals, JFIF YCbCr -> RGB.  Inverse DCT and colour conversion are the fixed-point forms of the IJG decoders
 * (LL&M factorisation with 13-bit constants; 16-bit colour constants), so a 4:4:4 file decodes to the bytes libjpeg /
 * libjpeg-turbo produce: tests/test_host_cpp.py compares against libjpeg (PIL) on synthetic images and, where the
 * reference is present, on its two textures (equal byte for byte).  zune-jpeg, the reference's decoder, is an
 * IEEE 1180-class integer IDCT as well, but it is not available here: PARITY UNPINNED against zune-jpeg itself.
 * Subsampled chroma is replicated (libjpeg smooths it by default).  Progressive files are refused with an error.
 *
 * Also reads binary PPM (P6), the format __graft_entry__.build() exports the pre-decoded textures in.
 */
#ifndef RAYTRACER_AMD_JPEG_DECODE_HPP
#define RAYTRACER_AMD_JPEG_DECODE_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace raytracer {

struct ImageError : std::runtime_error { /* image::ImageError */
    using std::runtime_error::runtime_error;
};

struct RgbImage; /* defined in raytracer.hpp: {width, height, rgb} */

namespace jpeg_detail {

struct Huff {
    /* canonical code tables, T.81 annex C / F.2.2.3 */
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
    void build(const uint8_t bits[16], const uint8_t* v, int nv) {
        memcpy(vals, v, (size_t)nv);
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l - 1];
            k += bits[l - 1];
            maxcode[l] = bits[l - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    int bw = 0, bh = 0; /* size in blocks, padded to whole MCUs */
    std::vector<uint8_t> plane;
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    int bit() {
        if (n == 0) {
            uint8_t b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    uint8_t b2 = p < end ? *p : 0;
                    if (b2 == 0x00) ++p; /* stuffed zero */
                    else { hit_marker = true; --p; b = 0; } /* marker: feed zeros until the caller handles it */
                }
            }
            acc = b;
            n = 8;
        }
        --n;
        return (int)((acc >> n) & 1u);
    }
    int bits(int c) {
        int v = 0;
        for (int i = 0; i < c; ++i) v = (v << 1) | bit();
        return v;
    }
    void reset() { n = 0; acc = 0; hit_marker = false; }
};

inline int decode_sym(BitReader& br, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    throw ImageError("jpeg: bad Huffman code");
}
inline int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; } /* T.81 F.2.2.1 */

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* 8x8 inverse DCT: the Loeffler-Ligtenberg-Moschytz factorisation (ICASSP 1989, 12 multiplies per 1-D pass) in the fixed-point
 * form every IJG-derived decoder uses - 13-bit constants, two extra bits kept between the passes, one rounding per pass - so that
 * the samples are the ones libjpeg / libjpeg-turbo ("islow") produce, not merely within +-1 of them: a +-1 difference per
 * component grows to +-4 in RGB through the colour matrix, which is what round 1's floating-point IDCT showed against libjpeg. */
inline int descale(long x, int n) { return (int)((x + (1L << (n - 1))) >> n); }
inline void idct_1d(const long in[8], long out[8]) { /* in: 8 coefficients; out: 8 samples scaled by 2^13 */
    const long F0_298631336 = 2446, F0_390180644 = 3196, F0_541196100 = 4433, F0_765366865 = 6270, F0_899976223 = 7373, F1_175875602 = 9633,
               F1_501321110 = 12299, F1_847759065 = 15137, F1_961570560 = 16069, F2_053119869 = 16819, F2_562915447 = 20995, F3_072711026 = 25172;
    /* even part */
    long z1 = (in[2] + in[6]) * F0_541196100;
    const long e2 = z1 - in[6] * F1_847759065, e3 = z1 + in[2] * F0_765366865;
    const long e0 = (in[0] + in[4]) * 8192, e1 = (in[0] - in[4]) * 8192;
    const long t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
    /* odd part */
    long o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
    z1 = o0 + o3;
    long z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
    const long z5 = (z3 + z4) * F1_175875602;
    o0 *= F0_298631336; o1 *= F2_053119869; o2 *= F3_072711026; o3 *= F1_501321110;
    z1 *= -F0_899976223; z2 *= -F2_562915447; z3 *= -F1_961570560; z4 *= -F0_390180644;
    z3 += z5; z4 += z5;
    o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
    out[0] = t10 + o3; out[7] = t10 - o3;
    out[1] = t11 + o2; out[6] = t11 - o2;
    out[2] = t12 + o1; out[5] = t12 - o1;
    out[3] = t13 + o0; out[4] = t13 - o0;
}
inline void idct_block(const int coef[64], uint8_t* out, int stride) {
    long ws[64];
    for (int x = 0; x < 8; ++x) { /* pass 1: columns, results keep 2 fractional bits */
        long in[8], o[8];
        for (int v = 0; v < 8; ++v) in[v] = coef[v * 8 + x];
        idct_1d(in, o);
        for (int v = 0; v < 8; ++v) ws[v * 8 + x] = descale(o[v], 13 - 2);
    }
    for (int y = 0; y < 8; ++y) { /* pass 2: rows, remove the 2 bits and the factor 8 of the 2-D transform, level shift, clamp */
        long o[8];
        idct_1d(ws + y * 8, o);
        for (int x = 0; x < 8; ++x) {
            const int r = descale(o[x], 13 + 2 + 3) + 128;
            out[y * stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}

inline uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

} /* namespace jpeg_detail */

/* Decodes a baseline JPEG held in memory to interleaved RGB8. */
inline void decode_jpeg(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgb) {
    using namespace jpeg_detail;
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) throw ImageError("jpeg: missing SOI");
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int W = 0, H = 0, restart = 0, hmax = 1, vmax = 1;
    bool have_sof = false, adobe_rgb = false;
    size_t pos = 2;
    for (;;) {
        if (pos + 4 > size) throw ImageError("jpeg: truncated before SOS");
        if (data[pos] != 0xFF) throw ImageError("jpeg: marker expected");
        while (pos < size && data[pos] == 0xFF) ++pos; /* fill bytes */
        uint8_t m = data[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) throw ImageError("jpeg: EOI before SOS");
        if (pos + 2 > size) throw ImageError("jpeg: truncated segment");
        size_t len = be16(data + pos);
        if (len < 2 || pos + len > size) throw ImageError("jpeg: bad segment length");
        const uint8_t* s = data + pos + 2;
        const uint8_t* e = data + pos + len;
        if (m == 0xDB) { /* DQT */
            while (s < e) {
                int pq = s[0] >> 4, tq = s[0] & 15;
                ++s;
                if (tq > 3 || s + (pq ? 128 : 64) > e) throw ImageError("jpeg: bad DQT");
                for (int i = 0; i < 64; ++i) { qt[tq][ZIGZAG[i]] = pq ? be16(s) : *s; s += pq ? 2 : 1; }
                qt_ok[tq] = true;
            }
        } else if (m == 0xC0 || m == 0xC1) { /* SOF0 / SOF1 */
            if (len < 8 || s[0] != 8) throw ImageError("jpeg: only 8-bit precision is supported");
            H = be16(s + 1);
            W = be16(s + 3);
            int nc = s[5];
            if (W <= 0 || H <= 0 || (nc != 1 && nc != 3) || (int)len < 8 + 3 * nc) throw ImageError("jpeg: unsupported frame header");
            comps.assign((size_t)nc, Component());
            for (int i = 0; i < nc; ++i) {
                comps[i].id = s[6 + 3 * i];
                comps[i].h = s[7 + 3 * i] >> 4;
                comps[i].v = s[7 + 3 * i] & 15;
                comps[i].tq = s[8 + 3 * i];
                if (comps[i].h < 1 || comps[i].h > 2 || comps[i].v < 1 || comps[i].v > 2 || comps[i].tq > 3)
                    throw ImageError("jpeg: unsupported sampling factors");
                hmax = comps[i].h > hmax ? comps[i].h : hmax;
                vmax = comps[i].v > vmax ? comps[i].v : vmax;
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            throw ImageError("jpeg: only baseline / sequential Huffman files are supported (progressive, lossless or arithmetic found)");
        } else if (m == 0xC4) { /* DHT */
            while (s < e) {
                int tc = s[0] >> 4, th = s[0] & 15;
                if (tc > 1 || th > 3 || s + 17 > e) throw ImageError("jpeg: bad DHT");
                int nv = 0;
                for (int i = 0; i < 16; ++i) nv += s[1 + i];
                if (nv > 256 || s + 17 + nv > e) throw ImageError("jpeg: bad DHT");
                (tc ? ac[th] : dc[th]).build(s + 1, s + 17, nv);
                s += 17 + nv;
            }
        } else if (m == 0xDD) { /* DRI */
            if (len >= 4) restart = be16(s);
        } else if (m == 0xEE) { /* APP14 Adobe: transform 0 with three components = RGB stored directly */
            if (len >= 14 && memcmp(s, "Adobe", 5) == 0 && s[11] == 0) adobe_rgb = true;
        } else if (m == 0xDA) { /* SOS: one interleaved scan with all components */
            if (!have_sof) throw ImageError("jpeg: SOS before SOF");
            int ns = s[0];
            if (ns != (int)comps.size() || (int)len < 6 + 2 * ns) throw ImageError("jpeg: non-interleaved scans are not supported");
            for (int i = 0; i < ns; ++i) {
                int cid = s[1 + 2 * i], k = -1;
                for (size_t c = 0; c < comps.size(); ++c) if (comps[c].id == cid) k = (int)c;
                if (k < 0) throw ImageError("jpeg: scan names an unknown component");
                comps[k].td = s[2 + 2 * i] >> 4;
                comps[k].ta = s[2 + 2 * i] & 15;
                if (comps[k].td > 3 || comps[k].ta > 3 || !dc[comps[k].td].present || !ac[comps[k].ta].present || !qt_ok[comps[k].tq])
                    throw ImageError("jpeg: scan uses an undefined table");
            }
            pos += len;
            break;
        }
        pos += len;
    }
    /* ---- entropy-coded data ---- */
    const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    const int mx = (W + mcu_w - 1) / mcu_w, my = (H + mcu_h - 1) / mcu_h;
    for (auto& c : comps) {
        c.bw = mx * c.h;
        c.bh = my * c.v;
        c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
    }
    BitReader br{data + pos, data + size};
    int until_restart = restart;
    for (int my_i = 0; my_i < my; ++my_i)
        for (int mx_i = 0; mx_i < mx; ++mx_i) {
            if (restart && until_restart == 0) { /* RSTn: byte-align, skip the marker, reset the predictors */
                br.reset();
                while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                if (br.p + 1 < br.end) br.p += 2;
                for (auto& c : comps) c.pred = 0;
                until_restart = restart;
            }
            for (auto& c : comps)
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        int coef[64] = {0};
                        int t = decode_sym(br, dc[c.td]);
                        if (t > 11) throw ImageError("jpeg: bad DC category");
                        c.pred += extend(br.bits(t), t);
                        coef[0] = c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            int rs = decode_sym(br, ac[c.ta]);
                            int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r != 15) break; /* EOB */
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) throw ImageError("jpeg: AC run past the block");
                            coef[ZIGZAG[k]] = extend(br.bits(sz), sz) * qt[c.tq][ZIGZAG[k]];
                            ++k;
                        }
                        const int px = (mx_i * c.h + bx) * 8, py = (my_i * c.v + by) * 8;
                        idct_block(coef, c.plane.data() + (size_t)py * c.bw * 8 + px, c.bw * 8);
                    }
            if (restart) --until_restart;
        }
    /* ---- upsample (replicate) + colour ---- */
    width = (uint32_t)W;
    height = (uint32_t)H;
    rgb.resize((size_t)W * H * 3);
    auto sample = [&](const Component& c, int x, int y) {
        const int sx = x * c.h / hmax, sy = y * c.v / vmax;
        return (int)c.plane[(size_t)sy * c.bw * 8 + sx];
    };
    /* JFIF YCbCr -> RGB (ITU-R BT.601, full range) in 16-bit fixed point with one rounding per channel, as the IJG decoders do:
     * R = Y + 1.402 Cr, G = Y - 0.34414 Cb - 0.71414 Cr, B = Y + 1.772 Cb */
    auto fix = [](double x) { return (long)(x * 65536.0 + 0.5); };
    const long K_CR_R = fix(1.40200), K_CB_B = fix(1.77200), K_CR_G = fix(0.71414), K_CB_G = fix(0.34414), HALF = 1L << 15;
    auto clamp8 = [](long r) { return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); };
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t* o = rgb.data() + ((size_t)y * W + x) * 3;
            if (comps.size() == 1) {
                o[0] = o[1] = o[2] = (uint8_t)sample(comps[0], x, y);
            } else if (adobe_rgb) {
                o[0] = (uint8_t)sample(comps[0], x, y); o[1] = (uint8_t)sample(comps[1], x, y); o[2] = (uint8_t)sample(comps[2], x, y);
            } else { /* JFIF: ITU-R BT.601 full range */
                const long Y = sample(comps[0], x, y), cb = sample(comps[1], x, y) - 128, cr = sample(comps[2], x, y) - 128;
                o[0] = clamp8(Y + ((K_CR_R * cr + HALF) >> 16));
                o[1] = clamp8(Y + ((-K_CB_G * cb + HALF - K_CR_G * cr) >> 16));
                o[2] = clamp8(Y + ((K_CB_B * cb + HALF) >> 16));
            }
        }
}

/* binary PPM (P6, maxval 255) */
inline void decode_ppm(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgb) {
    size_t pos = 0;
    auto token = [&]() {
        for (;;) {
            while (pos < size && (data[pos] == ' ' || data[pos] == '\n' || data[pos] == '\r' || data[pos] == '\t')) ++pos;
            if (pos < size && data[pos] == '#') { while (pos < size && data[pos] != '\n') ++pos; continue; }
            break;
        }
        std::string t;
        while (pos < size && data[pos] > ' ') t.push_back((char)data[pos++]);
        return t;
    };
    if (token() != "P6") throw ImageError("ppm: not a P6 file");
    long w = std::atol(token().c_str()), h = std::atol(token().c_str()), mv = std::atol(token().c_str());
    ++pos; /* the single whitespace after maxval */
    if (w <= 0 || h <= 0 || mv != 255 || pos + (size_t)w * h * 3 > size) throw ImageError("ppm: bad header or truncated data");
    width = (uint32_t)w;
    height = (uint32_t)h;
    rgb.assign(data + pos, data + pos + (size_t)w * h * 3);
}

inline std::vector<uint8_t> read_file(const std::string& filename) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) throw ImageError("failed to open " + filename);
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    return buf;
}

/* image_loader.rs:5-8: file -> (width, height, RGB8).  Format by content: JPEG (FF D8) or binary PPM (P6). */
inline void load_image_to_rgb(const std::string& filename, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgb) {
    std::vector<uint8_t> buf = read_file(filename);
    if (buf.size() >= 2 && buf[0] == 0xFF && buf[1] == 0xD8) decode_jpeg(buf.data(), buf.size(), width, height, rgb);
    else if (buf.size() >= 2 && buf[0] == 'P' && buf[1] == '6') decode_ppm(buf.data(), buf.size(), width, height, rgb);
    else throw ImageError("unsupported image format: " + filename);
}

} /* namespace raytracer */
#endif
