/* jpeg_decode.c -- minimal baseline (SOF0, 8-bit, Huffman, no chroma subsampling) JPEG decoder to binary PPM.
 *
 * Development tool only: tests/golden/make_reference_fixtures.py uses it to read the texture images next to the reference's
 * test/textures project (the container has no image library) before shrinking them into small PNG fixtures.
 *   gcc -O2 -o /tmp/jpeg_decode tools/jpeg_decode.c -lm && /tmp/jpeg_decode in.jpg out.ppm
 * ITU T.81: Huffman tables (DHT), quantisation tables (DQT), restart intervals (DRI), JFIF YCbCr -> RGB. */
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
    fprintf(stderr, "bad huffman code\n");
    exit(2);
}
static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

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

int main(int argc, char** argv) {
    if (argc != 3) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    size = ftell(f);
    fseek(f, 0, SEEK_SET);
    data = malloc(size);
    if (fread(data, 1, size, f) != size) return 1;
    fclose(f);
    uint16_t qt[4][64];
    Huff dc[4], ac[4];
    int width = 0, height = 0, ncomp = 0, comp_q[4] = {0}, comp_dc[4] = {0}, comp_ac[4] = {0}, restart = 0;
    pos = 2;
    for (;;) {
        if (pos + 4 > size || data[pos] != 0xFF) return 3;
        int marker = data[pos + 1];
        int len = (data[pos + 2] << 8) | data[pos + 3];
        uint8_t* p = data + pos + 4;
        if (marker == 0xDB) {
            uint8_t* end = data + pos + 2 + len;
            while (p < end) {
                int pq = p[0] >> 4, tq = p[0] & 15;
                ++p;
                for (int i = 0; i < 64; ++i) {
                    qt[tq][zigzag[i]] = pq ? (uint16_t)((p[0] << 8) | p[1]) : p[0];
                    p += pq ? 2 : 1;
                }
            }
        } else if (marker == 0xC4) {
            uint8_t* end = data + pos + 2 + len;
            while (p < end) {
                int tc = p[0] >> 4, th = p[0] & 15;
                Huff* h = tc ? &ac[th] : &dc[th];
                int n = 0;
                h->bits[0] = 0;
                for (int i = 1; i <= 16; ++i) n += (h->bits[i] = p[i]);
                memcpy(h->vals, p + 17, n);
                build(h);
                p += 17 + n;
            }
        } else if (marker == 0xC0) {
            height = (p[1] << 8) | p[2];
            width = (p[3] << 8) | p[4];
            ncomp = p[5];
            for (int i = 0; i < ncomp; ++i) {
                if (p[7 + 3 * i] != 0x11) {
                    fprintf(stderr, "chroma subsampling is not supported\n");
                    return 4;
                }
                comp_q[i] = p[8 + 3 * i];
            }
        } else if (marker == 0xC2 || marker == 0xC1) {
            fprintf(stderr, "only baseline JPEG is supported\n");
            return 4;
        } else if (marker == 0xDD) {
            restart = (p[0] << 8) | p[1];
        } else if (marker == 0xDA) {
            int ns = p[0];
            for (int i = 0; i < ns; ++i) {
                comp_dc[i] = p[2 + 2 * i] >> 4;
                comp_ac[i] = p[2 + 2 * i] & 15;
            }
            pos += 2 + len;
            break;
        }
        pos += 2 + len;
    }
    int bw = (width + 7) / 8, bh = (height + 7) / 8;
    int W = bw * 8, H = bh * 8;
    uint8_t* planes[3];
    for (int i = 0; i < 3; ++i) planes[i] = calloc((size_t)W * H, 1);
    int pred[3] = {0, 0, 0}, count = 0;
    bitcnt = 0;
    for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
            if (restart && count == restart) {
                bitcnt = 0;
                while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7)) pos++;
                pos += 2;
                pred[0] = pred[1] = pred[2] = 0;
                count = 0;
            }
            for (int c = 0; c < ncomp; ++c) {
                int coef[64] = {0};
                int t = decode(&dc[comp_dc[c]]);
                int diff = t ? extend(getbits(t), t) : 0;
                pred[c] += diff;
                coef[0] = pred[c];
                for (int k = 1; k < 64;) {
                    int rs = decode(&ac[comp_ac[c]]);
                    int r = rs >> 4, s = rs & 15;
                    if (s == 0) {
                        if (r == 15) {
                            k += 16;
                            continue;
                        }
                        break;
                    }
                    k += r;
                    if (k > 63) break;
                    coef[zigzag[k]] = extend(getbits(s), s);
                    ++k;
                }
                idct(coef, qt[comp_q[c]], planes[c < 3 ? c : 2] + (size_t)by * 8 * W + bx * 8, W);
            }
            ++count;
        }
    FILE* o = fopen(argv[2], "wb");
    fprintf(o, "P6\n%d %d\n255\n", width, height);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            double Y = planes[0][(size_t)y * W + x], cb = ncomp > 1 ? planes[1][(size_t)y * W + x] - 128.0 : 0, cr = ncomp > 2 ? planes[2][(size_t)y * W + x] - 128.0 : 0;
            double rgb[3] = {Y + 1.402 * cr, Y - 0.344136 * cb - 0.714136 * cr, Y + 1.772 * cb};
            for (int i = 0; i < 3; ++i) {
                int v = (int)floor(rgb[i] + 0.5);
                fputc(v < 0 ? 0 : (v > 255 ? 255 : v), o);
            }
        }
    fclose(o);
    return 0;
}
