// jpeg_writer.cpp -- Image::saveJPG (src/image.cpp:60-74): the screenshot of saveImage(true) (src/main.cpp:105-144) as a JPEG file.
//
// The reference hands its 8-bit RGB rows to stb_image_write's stbi_write_jpg(..., quality 90) (vendored under external/include,
// itself Jon Olick's public-domain jo_jpeg): a baseline sequential JFIF file, 4:4:4 (no chroma subsampling), the Annex-K quantisation
// tables scaled to the quality, the Annex-K Huffman tables, an AAN float DCT, padding blocks at the right / bottom edge by repeating
// the last column / row.  A decoder shows the same picture for any conforming encoder, but the FILE is only the reference's file if
// every coefficient comes out the same, so this writer evaluates the same float expressions in the same order (colour transform,
// the eight-point AAN butterfly, scaling by 1 / (q * aan[row] * aan[col]), rounding half away from zero by truncation) and packs
// the same bit stream.  The Huffman code words are generated here from the (counts, symbols) form of the Annex-K tables.
// Pinned byte for byte against the reference's compiled Image::saveJPG (oracle/_ref/libref_loaders.so) on 40 pictures,
// tests/golden/jpeg_ref.npz, tests/test_host_and_abi.py.
#include <cstdio>
#include <cstring>
#include <vector>

#include "rs_internal.h"

namespace {

// ITU-T T.81 Annex K.3: BITS (codes per length 1..16) and HUFFVAL of the four typical tables
const unsigned char kDcLumBits[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
const unsigned char kDcChrBits[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
const unsigned char kDcVals[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
const unsigned char kAcLumBits[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
const unsigned char kAcChrBits[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
const unsigned char kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
    0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };
const unsigned char kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
    0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };
// Annex K.1 / K.2 quantisation tables in natural (row-major) order
const int kLumQ[64] = { 16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                        18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
const int kChrQ[64] = { 17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                        99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };

struct Code { unsigned short bits, length; };

// canonical code words of a (BITS, HUFFVAL) table (T.81 Annex C), indexed by symbol
void canonical_codes(const unsigned char bits[16], const unsigned char* vals, Code out[256]) {
    std::memset(out, 0, 256 * sizeof(Code));
    unsigned code = 0;
    int k = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < bits[len - 1]; i++, k++) { out[vals[k]].bits = (unsigned short)code; out[vals[k]].length = (unsigned short)len; code++; }
        code <<= 1;
    }
}

struct Stream {
    std::vector<unsigned char> bytes;
    int acc = 0, count = 0;               // pending bits, left-aligned in the low 24 bits of acc
    void put(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; bytes.insert(bytes.end(), b, b + n); }
    void put1(int v) { bytes.push_back((unsigned char)v); }
    void bits(Code c) {
        count += c.length;
        acc |= (int)c.bits << (24 - count);
        while (count >= 8) {
            const unsigned char b = (unsigned char)((acc >> 16) & 255);
            bytes.push_back(b);
            if (b == 255) bytes.push_back(0);          // byte stuffing
            acc <<= 8;
            count -= 8;
        }
    }
};

// the zig-zag position of natural index i
int zigzag_of(int i) {
    static int table[64];
    static bool ready = false;
    if (!ready) {
        int x = 0, y = 0;
        for (int k = 0; k < 64; k++) {
            table[y * 8 + x] = k;
            if ((x + y) % 2 == 0) { if (x == 7) y++; else if (y == 0) x++; else { x++; y--; } }
            else { if (y == 7) x++; else if (x == 0) y++; else { x--; y++; } }
        }
        ready = true;
    }
    return table[i];
}

// eight-point forward DCT of Arai, Agui and Nakajima (unscaled outputs; the scale factors are folded into the quantisation), on
// the elements p[0], p[stride], ... -- the float sequence of the reference's writer
void aan8(float* p, int stride) {
    const float a0 = p[0], a1 = p[stride], a2 = p[2 * stride], a3 = p[3 * stride], a4 = p[4 * stride], a5 = p[5 * stride], a6 = p[6 * stride], a7 = p[7 * stride];
    const float s07 = a0 + a7, d07 = a0 - a7, s16 = a1 + a6, d16 = a1 - a6, s25 = a2 + a5, d25 = a2 - a5, s34 = a3 + a4, d34 = a3 - a4;
    // even half
    const float e0 = s07 + s34, e3 = s07 - s34, e1 = s16 + s25, e2 = s16 - s25;
    p[0] = e0 + e1;
    p[4 * stride] = e0 - e1;
    const float r = (e2 + e3) * 0.707106781f;
    p[2 * stride] = e3 + r;
    p[6 * stride] = e3 - r;
    // odd half
    const float o0 = d34 + d25, o1 = d25 + d16, o2 = d16 + d07;
    const float z5 = (o0 - o2) * 0.382683433f;
    const float z2 = o0 * 0.541196100f + z5;
    const float z4 = o2 * 1.306562965f + z5;
    const float z3 = o1 * 0.707106781f;
    const float z11 = d07 + z3, z13 = d07 - z3;
    p[5 * stride] = z13 + z2;
    p[3 * stride] = z13 - z2;
    p[stride] = z11 + z4;
    p[7 * stride] = z11 - z4;
}

// magnitude category and the appended bits of a coefficient (T.81 F.1.2.1): negative values as value - 1 in `length` bits
Code amplitude(int v) {
    int mag = v < 0 ? -v : v, length = 1;
    while (mag >>= 1) length++;
    const int low = v < 0 ? v - 1 : v;
    return Code{ (unsigned short)(low & ((1 << length) - 1)), (unsigned short)length };
}

// one 8x8 block: DCT, quantise into zig-zag order, entropy-code; returns the block's DC value (the next block's predictor)
int encode_block(Stream& s, float* block, const float* scale, int dcPred, const Code* dcCodes, const Code* acCodes) {
    for (int r = 0; r < 8; r++) aan8(block + 8 * r, 1);
    for (int c = 0; c < 8; c++) aan8(block + c, 8);
    int q[64];
    for (int i = 0; i < 64; i++) {
        const float v = block[i] * scale[i];
        q[zigzag_of(i)] = (int)(v < 0 ? v - 0.5f : v + 0.5f);
    }
    const int diff = q[0] - dcPred;
    if (diff == 0) s.bits(dcCodes[0]);
    else { const Code a = amplitude(diff); s.bits(dcCodes[a.length]); s.bits(a); }
    int last = 63;
    while (last > 0 && q[last] == 0) last--;
    if (last == 0) { s.bits(acCodes[0x00]); return q[0]; }            // end of block right after the DC term
    for (int i = 1; i <= last; i++) {
        int run = 0;
        while (q[i] == 0 && i <= last) { run++; i++; }
        for (; run >= 16; run -= 16) s.bits(acCodes[0xF0]);            // sixteen zeros
        const Code a = amplitude(q[i]);
        s.bits(acCodes[(run << 4) + a.length]);
        s.bits(a);
    }
    if (last != 63) s.bits(acCodes[0x00]);
    return q[0];
}

}  // namespace

extern "C" int rs_write_jpg(const char* path, const unsigned char* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0 || width > 65535 || height > 65535) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_write_jpg: bad argument");
    // quality 90 (src/image.cpp:71) -> table scale 200 - 2 * 90 = 20 %
    const int quality = 90, pct = quality < 50 ? 5000 / quality : 200 - quality * 2;
    unsigned char lumTable[64], chrTable[64];             // in zig-zag order, as the DQT segment stores them
    for (int i = 0; i < 64; i++) {
        const int y = (kLumQ[i] * pct + 50) / 100, c = (kChrQ[i] * pct + 50) / 100;
        lumTable[zigzag_of(i)] = (unsigned char)(y < 1 ? 1 : y > 255 ? 255 : y);
        chrTable[zigzag_of(i)] = (unsigned char)(c < 1 ? 1 : c > 255 ? 255 : c);
    }
    // AAN output scale per row / column (times 2 * sqrt 2: the DCT's 1/8 normalisation folded in), as float products
    static const float aan[8] = { 1.0f * 2.828427125f, 1.387039845f * 2.828427125f, 1.306562965f * 2.828427125f, 1.175875602f * 2.828427125f,
                                  1.0f * 2.828427125f, 0.785694958f * 2.828427125f, 0.541196100f * 2.828427125f, 0.275899379f * 2.828427125f };
    float lumScale[64], chrScale[64];
    for (int r = 0, k = 0; r < 8; r++)
        for (int c = 0; c < 8; c++, k++) {
            lumScale[k] = 1 / (lumTable[zigzag_of(k)] * aan[r] * aan[c]);
            chrScale[k] = 1 / (chrTable[zigzag_of(k)] * aan[r] * aan[c]);
        }
    Code dcLum[256], dcChr[256], acLum[256], acChr[256];
    canonical_codes(kDcLumBits, kDcVals, dcLum); canonical_codes(kDcChrBits, kDcVals, dcChr);
    canonical_codes(kAcLumBits, kAcLumVals, acLum); canonical_codes(kAcChrBits, kAcChrVals, acChr);

    Stream s;
    s.bytes.reserve((size_t)width * height / 2 + 1024);
    // SOI, APP0 (JFIF 1.1, aspect 1:1), DQT with both tables
    const unsigned char soiApp0Dqt[] = { 0xFF, 0xD8, 0xFF, 0xE0, 0, 0x10, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0xFF, 0xDB, 0, 0x84, 0 };
    s.put(soiApp0Dqt, sizeof soiApp0Dqt); s.put(lumTable, 64); s.put1(1); s.put(chrTable, 64);
    // SOF0: 8 bits, three components 1x1, Y on table 0, Cb / Cr on table 1; then DHT with the four tables
    const unsigned char sof0Dht[] = { 0xFF, 0xC0, 0, 0x11, 8, (unsigned char)(height >> 8), (unsigned char)(height & 0xff), (unsigned char)(width >> 8), (unsigned char)(width & 0xff),
                                      3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1, 0xFF, 0xC4, 0x01, 0xA2, 0 };
    s.put(sof0Dht, sizeof sof0Dht);
    s.put(kDcLumBits, 16); s.put(kDcVals, 12); s.put1(0x10);
    s.put(kAcLumBits, 16); s.put(kAcLumVals, 162); s.put1(1);
    s.put(kDcChrBits, 16); s.put(kDcVals, 12); s.put1(0x11);
    s.put(kAcChrBits, 16); s.put(kAcChrVals, 162);
    const unsigned char sos[] = { 0xFF, 0xDA, 0, 0xC, 3, 1, 0, 2, 0x11, 3, 0x11, 0, 0x3F, 0 };
    s.put(sos, sizeof sos);

    int dcY = 0, dcU = 0, dcV = 0;
    for (int by = 0; by < height; by += 8)
        for (int bx = 0; bx < width; bx += 8) {
            float Y[64], U[64], V[64];
            for (int r = 0, k = 0; r < 8; r++) {
                const int yy = by + r < height ? by + r : height - 1;          // the last row / column repeats beyond the edge
                for (int c = 0; c < 8; c++, k++) {
                    const int xx = bx + c < width ? bx + c : width - 1;
                    const unsigned char* p = rgb + ((size_t)yy * width + xx) * 3;
                    const float R = p[0], G = p[1], B = p[2];
                    Y[k] = +0.29900f * R + 0.58700f * G + 0.11400f * B - 128;
                    U[k] = -0.16874f * R - 0.33126f * G + 0.50000f * B;
                    V[k] = +0.50000f * R - 0.41869f * G - 0.08131f * B;
                }
            }
            dcY = encode_block(s, Y, lumScale, dcY, dcLum, acLum);
            dcU = encode_block(s, U, chrScale, dcU, dcChr, acChr);
            dcV = encode_block(s, V, chrScale, dcV, dcChr, acChr);
        }
    s.bits(Code{ 0x7F, 7 });                               // pad the last byte with ones
    s.put1(0xFF); s.put1(0xD9);                            // EOI

    FILE* f = std::fopen(path, "wb");
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_write_jpg: cannot open the file for writing");
    const bool ok = std::fwrite(s.bytes.data(), 1, s.bytes.size(), f) == s.bytes.size();
    std::fclose(f);
    return ok ? 0 : rs_fail(RS_ERR_UNSUPPORTED, "rs_write_jpg: short write");
}
