// images.cpp -- texture ingest for the C++ host layer: what Texture::from_path does (pyrite/src/texture.rs:25-85, :174-295).
//
// The reference decodes with the `image` crate and converts with `palette` 0.7.2; neither is vendored, so -- exactly as in
// pyrite_amd/images.py, whose results this file reproduces bit for bit -- the conversions are restated from their published
// definitions: the sRGB transfer function (IEC 61966-2-1, evaluated in f64) for non-`linear` textures, component / max for
// linear ones and for alpha, the Y row of the sRGB matrix for colour -> mono. PNG (non-interlaced; 8 / 16 bit; grey, grey +
// alpha, RGB, RGBA, palette; 1 / 2 / 4 bit grey and palette) is read here with a small inflate; baseline JPEG through
// ../jpeg.c.
#include <cmath>
#include <cstring>
#include <fstream>

#include "pyrite_host.hpp"

extern "C" int pyr_jpeg_decode(const uint8_t* bytes, size_t nbytes, int* out_width, int* out_height, uint8_t** out_rgb, char* error, size_t error_size);
extern "C" void pyr_image_free(uint8_t* p);

namespace pyrite {
namespace {

std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ProjectError("could not load " + path + ": no such file");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// ---- inflate (RFC 1951) of a zlib stream (RFC 1950) ---------------------------------------------------------------------------
struct BitReader {
    const uint8_t* data;
    size_t size, pos = 0;
    uint32_t bitbuf = 0;
    int bitcount = 0;
    int bit() {
        if (bitcount == 0) {
            if (pos >= size) throw ProjectError("PNG: truncated image data");
            bitbuf = data[pos++];
            bitcount = 8;
        }
        const int b = bitbuf & 1;
        bitbuf >>= 1;
        --bitcount;
        return b;
    }
    uint32_t bits(int n) {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) v |= (uint32_t)bit() << i;
        return v;
    }
};
struct Huffman {
    uint16_t count[16] = {0}, symbol[288] = {0};
    void build(const uint8_t* lengths, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16] = {0};
        for (int len = 1; len < 16; ++len) offs[len] = offs[len - 1] + count[len - 1];
        for (int i = 0; i < n; ++i)
            if (lengths[i] != 0) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            code |= br.bit();
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw ProjectError("PNG: bad Huffman code");
    }
};

std::vector<uint8_t> inflate_zlib(const std::vector<uint8_t>& z) {
    if (z.size() < 2) throw ProjectError("PNG: empty image data");
    BitReader br{z.data() + 2, z.size() - 2};
    std::vector<uint8_t> out;
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (int last = 0; !last;) {
        last = br.bit();
        const uint32_t type = br.bits(2);
        if (type == 0) {
            br.bitcount = 0;
            if (br.pos + 4 > br.size) throw ProjectError("PNG: truncated stored block");
            const uint32_t len = br.data[br.pos] | (br.data[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.size) throw ProjectError("PNG: truncated stored block");
            out.insert(out.end(), br.data + br.pos, br.data + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) throw ProjectError("PNG: bad deflate block");
        Huffman lit, dist;
        if (type == 1) {
            uint8_t l[288];
            for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            lit.build(l, 288);
            uint8_t d[30];
            std::memset(d, 5, sizeof(d));
            dist.build(d, 30);
        } else {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t lengths[320] = {0};
            for (int i = 0; i < ncode; ++i) lengths[order[i]] = (uint8_t)br.bits(3);
            Huffman code;
            code.build(lengths, 19);
            uint8_t all[320] = {0};
            for (int i = 0; i < nlen + ndist;) {
                const int sym = code.decode(br);
                if (sym < 16) {
                    all[i++] = (uint8_t)sym;
                } else {
                    uint8_t value = 0;
                    int repeat;
                    if (sym == 16) {
                        if (i == 0) throw ProjectError("PNG: bad code lengths");
                        value = all[i - 1];
                        repeat = 3 + (int)br.bits(2);
                    } else if (sym == 17) {
                        repeat = 3 + (int)br.bits(3);
                    } else {
                        repeat = 11 + (int)br.bits(7);
                    }
                    if (i + repeat > nlen + ndist) throw ProjectError("PNG: bad code lengths");
                    while (repeat--) all[i++] = value;
                }
            }
            lit.build(all, nlen);
            dist.build(all + nlen, ndist);
        }
        for (;;) {
            const int sym = lit.decode(br);
            if (sym < 256) {
                out.push_back((uint8_t)sym);
            } else if (sym == 256) {
                break;
            } else {
                if (sym > 285) throw ProjectError("PNG: bad length symbol");
                const int len = len_base[sym - 257] + (int)br.bits(len_extra[sym - 257]);
                const int ds = dist.decode(br);
                if (ds > 29) throw ProjectError("PNG: bad distance symbol");
                const size_t d = dist_base[ds] + br.bits(dist_extra[ds]);
                if (d > out.size()) throw ProjectError("PNG: distance too far back");
                for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
            }
        }
    }
    return out;
}

struct Image { // integer samples as decoded: [height][width][channels], max = 255 or 65535
    uint32_t width = 0, height = 0, channels = 0, max = 255;
    std::vector<uint16_t> samples;
};

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

Image read_png(const std::string& path) {
    const std::vector<uint8_t> data = read_file(path);
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (data.size() < 8 || std::memcmp(data.data(), magic, 8) != 0) throw ProjectError(path + ": not a PNG file");
    uint32_t width = 0, height = 0;
    int depth = 0, color_type = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    for (size_t pos = 8; pos + 8 <= data.size();) {
        const uint32_t length = be32(&data[pos]);
        const std::string kind(reinterpret_cast<const char*>(&data[pos + 4]), 4);
        const uint8_t* body = &data[pos + 8];
        if (pos + 12 + length > data.size()) throw ProjectError(path + ": truncated PNG chunk");
        pos += 12 + length;
        if (kind == "IHDR") {
            width = be32(body), height = be32(body + 4);
            depth = body[8], color_type = body[9], interlace = body[12];
        } else if (kind == "PLTE") {
            palette.assign(body, body + length);
        } else if (kind == "tRNS") {
            trns.assign(body, body + length);
        } else if (kind == "IDAT") {
            idat.insert(idat.end(), body, body + length);
        } else if (kind == "IEND") {
            break;
        }
    }
    if (interlace) throw ProjectError(path + ": interlaced PNGs are not supported");
    uint32_t channels;
    switch (color_type) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw ProjectError(path + ": unknown PNG colour type");
    }
    if (depth != 8 && depth != 16 && !((color_type == 0 || color_type == 3) && (depth == 1 || depth == 2 || depth == 4)))
        throw ProjectError(path + ": unsupported bit depth " + std::to_string(depth));
    const uint32_t bits_per_pixel = channels * depth;
    const uint32_t bpp = std::max(1u, bits_per_pixel / 8);
    const size_t stride = ((size_t)width * bits_per_pixel + 7) / 8;
    const std::vector<uint8_t> raw = inflate_zlib(idat);
    if (raw.size() < (stride + 1) * height) throw ProjectError(path + ": truncated PNG image data");
    std::vector<uint8_t> out(stride * height), prev(stride, 0);
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t ftype = raw[y * (stride + 1)];
        const uint8_t* line = &raw[y * (stride + 1) + 1];
        uint8_t* cur = &out[y * stride];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (ftype) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            default: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            }
            cur[i] = (uint8_t)((line[i] + pred) & 0xFF);
        }
        std::memcpy(prev.data(), cur, stride);
    }
    Image img;
    img.width = width, img.height = height;
    std::vector<uint16_t> px; // [h][w][channels] before palette expansion
    if (depth == 16) {
        img.max = 65535;
        px.resize((size_t)width * height * channels);
        for (uint32_t y = 0; y < height; ++y)
            for (size_t i = 0; i < (size_t)width * channels; ++i) px[(size_t)y * width * channels + i] = (uint16_t)((out[y * stride + 2 * i] << 8) | out[y * stride + 2 * i + 1]);
    } else if (depth == 8) {
        px.resize((size_t)width * height * channels);
        for (uint32_t y = 0; y < height; ++y)
            for (size_t i = 0; i < (size_t)width * channels; ++i) px[(size_t)y * width * channels + i] = out[y * stride + i];
    } else { // packed 1 / 2 / 4-bit grey or palette indices
        px.resize((size_t)width * height);
        for (uint32_t y = 0; y < height; ++y)
            for (uint32_t x = 0; x < width; ++x) {
                const size_t bit = (size_t)x * depth;
                uint16_t v = (uint16_t)((out[y * stride + bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1));
                if (color_type == 0) v = (uint16_t)(v * 255 / ((1 << depth) - 1));
                px[(size_t)y * width + x] = v;
            }
    }
    if (color_type == 3) {
        const bool alpha = !trns.empty();
        img.channels = alpha ? 4 : 3;
        img.samples.resize((size_t)width * height * img.channels);
        for (size_t i = 0; i < (size_t)width * height; ++i) {
            const size_t k = px[i];
            if (3 * k + 2 >= palette.size()) throw ProjectError(path + ": palette index out of range");
            for (int c = 0; c < 3; ++c) img.samples[i * img.channels + c] = palette[3 * k + c];
            if (alpha) img.samples[i * 4 + 3] = k < trns.size() ? trns[k] : 255;
        }
    } else {
        img.channels = channels;
        img.samples = std::move(px);
    }
    return img;
}

Image read_jpeg(const std::string& path) {
    const std::vector<uint8_t> data = read_file(path);
    int w = 0, h = 0;
    uint8_t* rgb = nullptr;
    char err[256] = {0};
    if (pyr_jpeg_decode(data.data(), data.size(), &w, &h, &rgb, err, sizeof(err)) != 0) throw ProjectError(path + ": " + err);
    Image img;
    img.width = (uint32_t)w, img.height = (uint32_t)h, img.channels = 3;
    img.samples.assign(rgb, rgb + (size_t)w * h * 3);
    pyr_image_free(rgb);
    return img;
}

float srgb_to_linear(float c32) { // IEC 61966-2-1, evaluated in f64 and rounded to f32
    const double c = c32;
    return (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
}

} // namespace

// Texture::from_path (texture.rs:25-85) + convert_pixels (:174-199): image file -> linear f32 texels, [h][w][4] (LinSrgba) or
// [h][w] (LinLuma); the format is chosen by the file's extension (image::ImageFormat::from_path, texture.rs:32-35).
std::vector<float> load_texture_file(const std::string& path, bool linear, bool mono, uint32_t& width, uint32_t& height) {
    const size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    for (char& c : ext) c = (char)std::tolower((unsigned char)c);
    Image img;
    if (ext == ".png")
        img = read_png(path);
    else if (ext == ".jpg" || ext == ".jpeg")
        img = read_jpeg(path);
    else
        throw ProjectError(path + ": unsupported image format (PNG and baseline JPEG are read)");
    width = img.width, height = img.height;
    const uint32_t ch = img.channels;
    const bool has_alpha = ch == 2 || ch == 4;
    const uint32_t colors = has_alpha ? ch - 1 : ch;
    const float max = (float)img.max;
    const size_t n = (size_t)width * height;
    std::vector<float> out(mono ? n : n * 4);
    static const float luma_weights[3] = {0.2126729f, 0.7151522f, 0.0721750f}; // the Y row of the sRGB (D65) RGB -> XYZ matrix
    for (size_t i = 0; i < n; ++i) {
        float color[3];
        for (uint32_t c = 0; c < colors; ++c) {
            const float unit = (float)img.samples[i * ch + c] / max;
            color[c] = linear ? unit : srgb_to_linear(unit);
        }
        const float alpha = has_alpha ? (float)img.samples[i * ch + ch - 1] / max : 1.0f;
        if (mono) {
            out[i] = colors == 1 ? color[0] : color[0] * luma_weights[0] + color[1] * luma_weights[1] + color[2] * luma_weights[2];
        } else {
            if (colors == 1) color[1] = color[2] = color[0];
            out[4 * i] = color[0], out[4 * i + 1] = color[1], out[4 * i + 2] = color[2], out[4 * i + 3] = alpha;
        }
    }
    return out;
}

} // namespace pyrite
