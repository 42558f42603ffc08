/* jpeg.c -- baseline JPEG reader for texture ingest (libpyrite_images.so, host only, no HIP).
 *
 * The reference loads textures with the `image` crate (texture.rs:25-36); its own test project uses JPEG files
 * (pyrite/test/textures). This is a plain ITU T.81 baseline decoder: SOF0, 8-bit, Huffman, DQT / DHT / DRI, chroma
 * subsampling by pixel replication, JFIF YCbCr -> RGB, reference-accuracy floating-point IDCT. Progressive files and
 * baseline files whose scan does not interleave all components are rejected. Decoded values can differ from another
 * decoder's by a level or two (IDCT rounding, chroma upsampling filter).
 *
 * Texture files are untrusted input: every segment is parsed inside its declared length and inside the file, table and
 * component indices are checked, coefficient categories beyond baseline's are refused, allocations are checked and the
 * image size is capped. tests/test_textures.py runs truncated and bit-flipped files through an AddressSanitizer build. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint8_t bits[17];
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];
} Huff;

static uint8_t* data;
static size_t size, pos;
static uint32_t bitbuf;
static int bitcnt;

static void build(Huff* h) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->bits[l];
        k += h->bits[l];
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
}

static int getbit(void) {
    if (bitcnt == 0) {
        uint8_t b = pos < size ? data[pos++] : 0;
        if (b == 0xFF) {
            uint8_t n = pos < size ? data[pos] : 0;
            if (n == 0) pos++; /* stuffed byte */
        }
        bitbuf = b;
        bitcnt = 8;
    }
    bitcnt--;
    return (bitbuf >> bitcnt) & 1;
}
static int getbits(int n) {
    int v = 0;
    while (n--) v = (v << 1) | getbit();
    return v;
}
static int decode(const Huff* h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | getbit();
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->vals[h->valptr[l] + code - h->mincode[l]];
    }
    return -1;
}
static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; } /* 1 <= t <= 11 at every call */

static const int zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static void idct(const int* in, const uint16_t* q, uint8_t* out, int stride) {
    static double c[8][8];
    static int init = 0;
    if (!init) {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? sqrt(0.5) : 1.0) * cos((2 * x + 1) * u * M_PI / 16.0) * 0.5;
        init = 1;
    }
    double tmp[64], f[64];
    for (int i = 0; i < 64; ++i) f[i] = (double)in[i] * q[i];
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            double s = 0;
            for (int u = 0; u < 8; ++u) s += c[x][u] * f[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y) {
            double s = 0;
            for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
            int p = (int)floor(s + 128.5);
            out[y * stride + x] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));
        }
}

static int fail_with(const char* message, char* error, size_t error_size) {
    if (error && error_size) snprintf(error, error_size, "%s", message);
    return -1;
}

/* Decodes `size` bytes of a JPEG file. On success returns 0 and hands back a malloc'ed width * height * 3 RGB buffer
 * (release it with pyr_image_free); on failure returns -1 with a message in `error`. Not re-entrant (static state). */
int pyr_jpeg_decode(const uint8_t* bytes, size_t nbytes, int* out_width, int* out_height, uint8_t** out_rgb, char* error, size_t error_size) {
    if (!bytes || !out_width || !out_height || !out_rgb) return fail_with("null argument", error, error_size);
    data = (uint8_t*)bytes;
    size = nbytes;
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail_with("not a JPEG file", error, error_size);
    static uint16_t qt[4][64];
    static Huff dc[4], ac[4];
    int width = 0, height = 0, ncomp = 0, comp_q[4] = {0}, comp_dc[4] = {0}, comp_ac[4] = {0}, comp_h[4] = {1, 1, 1, 1}, comp_v[4] = {1, 1, 1, 1}, restart = 0;
    pos = 2;
    memset(qt, 0, sizeof(qt));
    memset(dc, 0, sizeof(dc));
    memset(ac, 0, sizeof(ac));
    for (;;) {
        if (pos + 4 > size || data[pos] != 0xFF) return fail_with("corrupt JPEG marker stream", error, error_size);
        int marker = data[pos + 1];
        if (marker == 0xFF) {
            pos++;
            continue;
        }
        if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD9)) return fail_with("corrupt JPEG marker stream", error, error_size); /* no length: not legal here */
        int len = (data[pos + 2] << 8) | data[pos + 3];
        if (len < 2 || pos + 2 + (size_t)len > size) return fail_with("truncated JPEG segment", error, error_size);
        const uint8_t* p = data + pos + 4;
        const uint8_t* end = data + pos + 2 + len; /* one past the segment's last byte */
        if (marker == 0xDB) {
            while (p < end) {
                int pq = p[0] >> 4, tq = p[0] & 15;
                ++p;
                if (pq > 1 || tq > 3 || (size_t)(end - p) < (pq ? 128u : 64u)) return fail_with("truncated or corrupt JPEG quantisation table", error, error_size);
                for (int i = 0; i < 64; ++i) {
                    qt[tq][zigzag[i]] = pq ? (uint16_t)((p[0] << 8) | p[1]) : p[0];
                    p += pq ? 2 : 1;
                }
            }
        } else if (marker == 0xC4) {
            while (p < end) {
                if ((size_t)(end - p) < 17) return fail_with("truncated JPEG Huffman table", error, error_size);
                int tc = p[0] >> 4, th = p[0] & 15;
                if (tc > 1 || th > 3) return fail_with("corrupt Huffman table", error, error_size);
                Huff* h = tc ? &ac[th] : &dc[th];
                int n = 0;
                h->bits[0] = 0;
                for (int i = 1; i <= 16; ++i) n += (h->bits[i] = p[i]);
                if (n > 256 || (size_t)(end - p) < 17u + (size_t)n) return fail_with("corrupt Huffman table", error, error_size);
                memcpy(h->vals, p + 17, (size_t)n);
                build(h);
                p += 17 + n;
            }
        } else if (marker == 0xC0) {
            if (len < 8) return fail_with("truncated JPEG frame header", error, error_size);
            height = (p[1] << 8) | p[2];
            width = (p[3] << 8) | p[4];
            ncomp = p[5];
            if (p[0] != 8 || (ncomp != 1 && ncomp != 3) || width <= 0 || height <= 0) return fail_with("unsupported JPEG frame", error, error_size);
            if (len < 8 + 3 * ncomp) return fail_with("truncated JPEG frame header", error, error_size);
            if ((uint64_t)width * (uint64_t)height > (1ull << 28)) return fail_with("JPEG image too large (more than 2^28 pixels)", error, error_size);
            for (int i = 0; i < ncomp; ++i) {
                comp_h[i] = p[7 + 3 * i] >> 4;
                comp_v[i] = p[7 + 3 * i] & 15;
                comp_q[i] = p[8 + 3 * i];
                if (comp_q[i] > 3) return fail_with("corrupt JPEG frame header", error, error_size);
                if (comp_h[i] < 1 || comp_h[i] > 2 || comp_v[i] < 1 || comp_v[i] > 2) return fail_with("unsupported JPEG sampling factors", error, error_size);
            }
        } else if (marker == 0xC2 || marker == 0xC1 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
            return fail_with("only baseline JPEG is supported (this file is progressive / extended / arithmetic coded)", error, error_size);
        } else if (marker == 0xDD) {
            if (len < 4) return fail_with("truncated JPEG segment", error, error_size);
            restart = (p[0] << 8) | p[1];
        } else if (marker == 0xDA) {
            if (width == 0) return fail_with("JPEG has no frame header", error, error_size);
            if (len < 3) return fail_with("truncated JPEG scan header", error, error_size);
            int ns = p[0];
            /* a baseline file may spread its components over several scans; this reader decodes one interleaved scan */
            if (ns != ncomp) return fail_with("unsupported JPEG: the scan does not interleave all components", error, error_size);
            if (len < 6 + 2 * ns) return fail_with("truncated JPEG scan header", error, error_size);
            for (int i = 0; i < ns; ++i) {
                comp_dc[i] = p[2 + 2 * i] >> 4;
                comp_ac[i] = p[2 + 2 * i] & 15;
                if (comp_dc[i] > 3 || comp_ac[i] > 3) return fail_with("corrupt JPEG scan header", error, error_size);
            }
            pos += 2 + (size_t)len;
            break;
        }
        pos += 2 + (size_t)len;
    }
    if (width == 0) return fail_with("JPEG has no frame header", error, error_size);
    int hmax = 1, vmax = 1;
    for (int i = 0; i < ncomp; ++i) {
        if (comp_h[i] > hmax) hmax = comp_h[i];
        if (comp_v[i] > vmax) vmax = comp_v[i];
    }
    const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
    uint8_t* planes[3] = {0, 0, 0};
    int plane_w[3] = {0, 0, 0};
    for (int i = 0; i < ncomp; ++i) {
        plane_w[i] = mcus_x * 8 * comp_h[i];
        planes[i] = (uint8_t*)calloc((size_t)plane_w[i] * mcus_y * 8 * comp_v[i], 1);
        if (!planes[i]) {
            for (int j = 0; j < i; ++j) free(planes[j]);
            return fail_with("out of memory decoding a JPEG", error, error_size);
        }
    }
    int pred[3] = {0, 0, 0}, count = 0;
    bitcnt = 0;
    for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
            if (restart && count == restart) {
                bitcnt = 0;
                while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7)) pos++;
                pos += 2;
                pred[0] = pred[1] = pred[2] = 0;
                count = 0;
            }
            for (int c = 0; c < ncomp; ++c)
                for (int v = 0; v < comp_v[c]; ++v)
                    for (int h = 0; h < comp_h[c]; ++h) {
                        int coef[64] = {0};
                        int t = decode(&dc[comp_dc[c]]);
                        if (t < 0 || t > 11) goto corrupt; /* baseline DC differences have at most 11 bits */
                        int diff = t ? extend(getbits(t), t) : 0;
                        pred[c] += diff;
                        coef[0] = pred[c];
                        for (int k = 1; k < 64;) {
                            int rs = decode(&ac[comp_ac[c]]);
                            if (rs < 0) goto corrupt;
                            int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r == 15) {
                                    k += 16;
                                    continue;
                                }
                                break;
                            }
                            if (s > 10) goto corrupt; /* baseline AC coefficients have at most 10 bits */
                            k += r;
                            if (k > 63) break;
                            coef[zigzag[k]] = extend(getbits(s), s);
                            ++k;
                        }
                        idct(coef, qt[comp_q[c]], planes[c] + (size_t)(my * comp_v[c] + v) * 8 * plane_w[c] + (mx * comp_h[c] + h) * 8, plane_w[c]);
                    }
            ++count;
        }
    {
        uint8_t* rgb = (uint8_t*)malloc((size_t)width * height * 3);
        if (!rgb) {
            for (int i = 0; i < ncomp; ++i) free(planes[i]);
            return fail_with("out of memory decoding a JPEG", error, error_size);
        }
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x) {
                double Y = planes[0][(size_t)(y * comp_v[0] / vmax) * plane_w[0] + x * comp_h[0] / hmax];
                double cb = 0, cr = 0;
                if (ncomp == 3) {
                    cb = planes[1][(size_t)(y * comp_v[1] / vmax) * plane_w[1] + x * comp_h[1] / hmax] - 128.0;
                    cr = planes[2][(size_t)(y * comp_v[2] / vmax) * plane_w[2] + x * comp_h[2] / hmax] - 128.0;
                }
                double c3[3] = {Y + 1.402 * cr, Y - 0.344136 * cb - 0.714136 * cr, Y + 1.772 * cb};
                for (int i = 0; i < 3; ++i) {
                    int v = (int)floor(c3[i] + 0.5);
                    rgb[((size_t)y * width + x) * 3 + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
            }
        for (int i = 0; i < ncomp; ++i) free(planes[i]);
        *out_width = width;
        *out_height = height;
        *out_rgb = rgb;
        return 0;
    }
corrupt:
    for (int i = 0; i < ncomp; ++i) free(planes[i]);
    return fail_with("corrupt JPEG entropy-coded data", error, error_size);
}

void pyr_image_free(uint8_t* p) { free(p); }
