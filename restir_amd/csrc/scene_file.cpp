// scene_file.cpp -- the reference's scene-file front end on the host:
//   Scene::Scene(filename)        src/scene.cpp:96-131      line-oriented text format (Material / Object / Camera / EnvMap)
//   Scene::loadMaterial           src/scene.cpp:371-433
//   Scene::loadModel              src/scene.cpp:222-283     OBJ file, material link, Translate / Rotate / Scale
//   Scene::loadCamera             src/scene.cpp:285-354
//   Resource::loadOBJMesh         src/scene.cpp:27-61       (tinyobj there; a reader of the same subset here)
//   instance baking of buildDevData   src/scene.cpp:161-176
//   Math::buildTransformationMatrix   src/mathUtil.cpp:13-20 + glm translate / rotate / scale / inverse
// What it does not do: decode GIF / PSD / PIC (stb_image in the reference).  Texture and environment-map files must be
// PNG, JPEG, TGA, BMP or binary PPM (8-bit values: stbi_loadf's LDR path is value / 255 with stbi_ldr_to_hdr_gamma(1), src/scene.cpp:97) or Radiance HDR
// (.hdr, RGBE, flat or run-length coded: mantissa * 2^(e - 136)); rows flipped for textures (stbi_set_flip_vertically_on_load(true),
// :98) and not for the environment map (:124-126).
// glTF (Resource::loadGLTFMesh) is not read either.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "rs_internal.h"

using namespace rs;

namespace {

// ---- GLM 0.9.6.3 matrix arithmetic in its operation order (column-major, c[col][row]) --------------------------
struct M4 { float c[4][4]; };
M4 identity() { M4 m; std::memset(&m, 0, sizeof m); m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.f; return m; }
M4 mul(const M4& a, const M4& b) {                                   // type_mat4x4.inl:686-704
    M4 r;
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 4; k++)
            r.c[j][k] = ((a.c[0][k] * b.c[j][0] + a.c[1][k] * b.c[j][1]) + a.c[2][k] * b.c[j][2]) + a.c[3][k] * b.c[j][3];
    return r;
}
M4 translate(const M4& m, f3 v) {                                     // gtc/matrix_transform.inl:40-49
    M4 r = m;
    for (int k = 0; k < 4; k++) r.c[3][k] = ((m.c[0][k] * v.x + m.c[1][k] * v.y) + m.c[2][k] * v.z) + m.c[3][k];
    return r;
}
M4 rotate(const M4& m, float angle, f3 v) {                           // :52-85
    const float c = cosf(angle), s = sinf(angle);
    const f3 axis = normalize(v);
    const f3 temp = axis * (1.f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = 0 + temp.x * axis.y + s * axis.z;
    R[0][2] = 0 + temp.x * axis.z - s * axis.y;
    R[1][0] = 0 + temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = 0 + temp.y * axis.z + s * axis.x;
    R[2][0] = 0 + temp.z * axis.x + s * axis.y;
    R[2][1] = 0 + temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    M4 r;
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 4; k++) r.c[j][k] = (m.c[0][k] * R[j][0] + m.c[1][k] * R[j][1]) + m.c[2][k] * R[j][2];
    for (int k = 0; k < 4; k++) r.c[3][k] = m.c[3][k];
    return r;
}
M4 scale(const M4& m, f3 v) {                                         // :122-134
    M4 r;
    for (int k = 0; k < 4; k++) { r.c[0][k] = m.c[0][k] * v.x; r.c[1][k] = m.c[1][k] * v.y; r.c[2][k] = m.c[2][k] * v.z; r.c[3][k] = m.c[3][k]; }
    return r;
}
M4 build_transformation_matrix(f3 translation, f3 rotation, f3 scl) {  // mathUtil.cpp:13-20
    const M4 translationMat = translate(identity(), translation);
    M4 rotationMat = rotate(identity(), rotation.x * kPi / 180.f, mk3(1.f, 0.f, 0.f));
    rotationMat = mul(rotationMat, rotate(identity(), rotation.y * kPi / 180.f, mk3(0.f, 1.f, 0.f)));
    rotationMat = mul(rotationMat, rotate(identity(), rotation.z * kPi / 180.f, mk3(0.f, 0.f, 1.f)));
    const M4 scaleMat = scale(identity(), scl);
    return mul(mul(translationMat, rotationMat), scaleMat);
}
// glm::inverse of a 4x4 (type_mat4x4.inl:37-92) as the cofactor expansion it is: the eighteen 2x2 minors built from two rows and
// two of the columns 1..3, combined per output column with the entries of the remaining rows, alternating signs, divided by the
// expansion of the determinant along the first column's row -- with GLM's grouping of every sum, so the floats are GLM's.
M4 inverse(const M4& m) {
    static const int rowPair[6][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };
    static const int colPair[3][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 } };
    float minor[6][3];
    for (int g = 0; g < 6; g++)
        for (int k = 0; k < 3; k++) {
            const int ca = colPair[k][0], cb = colPair[k][1], ra = rowPair[g][0], rb = rowPair[g][1];
            minor[g][k] = m.c[ca][ra] * m.c[cb][rb] - m.c[cb][ra] * m.c[ca][rb];
        }
    // per lane k of the four-wide vectors GLM works with: the minor of group g (lanes 0 and 1 share the first) and row r of the
    // column that is left (column 1 for lane 0, column 0 otherwise)
    auto fac = [&](int g, int k) { return minor[g][k < 2 ? 0 : k - 1]; };
    auto vec = [&](int r, int k) { return m.c[k == 0 ? 1 : 0][r]; };
    M4 inv;
    for (int k = 0; k < 4; k++) {
        const float plus = (k & 1) ? -1.f : 1.f, minus = -plus;
        inv.c[0][k] = ((vec(1, k) * fac(0, k) - vec(2, k) * fac(1, k)) + vec(3, k) * fac(2, k)) * plus;
        inv.c[1][k] = ((vec(0, k) * fac(0, k) - vec(2, k) * fac(3, k)) + vec(3, k) * fac(4, k)) * minus;
        inv.c[2][k] = ((vec(0, k) * fac(1, k) - vec(1, k) * fac(3, k)) + vec(3, k) * fac(5, k)) * plus;
        inv.c[3][k] = ((vec(0, k) * fac(2, k) - vec(1, k) * fac(4, k)) + vec(2, k) * fac(5, k)) * minus;
    }
    const float det = (m.c[0][0] * inv.c[0][0] + m.c[0][1] * inv.c[1][0]) + (m.c[0][2] * inv.c[2][0] + m.c[0][3] * inv.c[3][0]);
    const float oneOverDet = 1.f / det;
    for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) inv.c[j][k] = inv.c[j][k] * oneOverDet;
    return inv;
}

// scene.cpp:169-170 with transform / normalMat of loadModel (:273-278)
void bake(const M4& tr, int n, const float* vertsIn, const float* normalsIn, float* vertsOut, float* normalsOut) {
    const M4 inv = inverse(tr);
    float nm[3][3];                                                   // transpose(mat3(transfInv)), columns
    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) nm[j][k] = inv.c[k][j];
    for (int i = 0; i < n; i++) {
        const f3 v = ld3(vertsIn + (size_t)i * 3), nn = ld3(normalsIn + (size_t)i * 3);
        f3 o;                                                         // mat4 * vec4 (type_mat4x4.inl:612-628)
        o.x = (tr.c[0][0] * v.x + tr.c[1][0] * v.y) + (tr.c[2][0] * v.z + tr.c[3][0] * 1.f);
        o.y = (tr.c[0][1] * v.x + tr.c[1][1] * v.y) + (tr.c[2][1] * v.z + tr.c[3][1] * 1.f);
        o.z = (tr.c[0][2] * v.x + tr.c[1][2] * v.y) + (tr.c[2][2] * v.z + tr.c[3][2] * 1.f);
        st3(vertsOut + (size_t)i * 3, o);
        const f3 q = mk3(nm[0][0] * nn.x + nm[1][0] * nn.y + nm[2][0] * nn.z,
                         nm[0][1] * nn.x + nm[1][1] * nn.y + nm[2][1] * nn.z,
                         nm[0][2] * nn.x + nm[1][2] * nn.y + nm[2][2] * nn.z);
        st3(normalsOut + (size_t)i * 3, normalize(q));
    }
}

// ---- text input ----------------------------------------------------------------------------------------------------
// The reference reads its files with utilityCore::safeGetline on an ifstream and splits lines with tokenizeString
// (src/utilities.cpp:57-95).  Here the file is read whole and walked in memory; what has to match is observable behaviour:
// a line ends at "\n", "\r\n" or a lone "\r"; a last line without terminator is still returned; `good()` turns false only
// when a read finds nothing at all left (so a file ending in a newline yields one more, empty, line first); tokens are maximal
// runs of non-whitespace as operator>> sees them (space, \t, \v, \f and stray \r / \n).
class Lines {
public:
    bool open(const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) text_.append(buf, got);
        std::fclose(f);
        return true;
    }
    bool good() const { return !exhausted_; }
    size_t size() const { return text_.size(); }
    // the same line as next(), as a range inside the file buffer (no copy)
    void next_span(const char*& b, const char*& e) {
        b = e = text_.data() + at_;
        if (exhausted_) return;
        size_t end = at_;
        while (end < text_.size() && text_[end] != '\n' && text_[end] != '\r') end++;
        e = text_.data() + end;
        if (end == text_.size()) { if (b == e) exhausted_ = true; at_ = end; }
        else at_ = end + ((text_[end] == '\r' && end + 1 < text_.size() && text_[end + 1] == '\n') ? 2 : 1);
    }
    const std::string& next() {
        line_.clear();
        if (exhausted_) return line_;                                  // reads on a failed stream return nothing
        size_t end = at_;
        while (end < text_.size() && text_[end] != '\n' && text_[end] != '\r') end++;
        line_.assign(text_, at_, end - at_);
        if (end == text_.size()) { if (line_.empty()) exhausted_ = true; at_ = end; }
        else at_ = end + ((text_[end] == '\r' && end + 1 < text_.size() && text_[end + 1] == '\n') ? 2 : 1);
        return line_;
    }
private:
    std::string text_, line_;
    size_t at_ = 0;
    bool exhausted_ = false;
};

std::vector<std::string> tokenize(const std::string& str) {
    std::vector<std::string> out;
    size_t i = 0;
    const auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
    while (i < str.size()) {
        while (i < str.size() && blank(str[i])) i++;
        size_t j = i;
        while (j < str.size() && !blank(str[j])) j++;
        if (j > i) out.emplace_back(str, i, j - i);
        i = j;
    }
    return out;
}

struct Mesh { std::vector<float> v, n, t; };                          // de-indexed: 3 / 3 / 2 floats per corner

// Number syntax of the OBJ reader the reference links (tinyobjloader 2.0, external/include/tiny_obj_loader.h:866-1010,
// "tryParseDouble"): digits are accumulated in double -- integer part by *10 + d, fraction digit k by d * 10^-k, a decimal
// exponent e as ldexp(m * 5^e, e) -- and the result is narrowed to float.  It is not strtof: for long decimals the two can
// differ in the last float bit, so the same accumulation is done here.  Unparsable text reads as 0.
float obj_real(const char* c, const char* end) {
    // 10^-k exactly as std::pow(10.0, -k) returns it (the reader calls it per digit; a table of the same values is faster)
    static const std::vector<double> tenTo = [] { std::vector<double> t(64); for (int k = 0; k < 64; k++) t[k] = std::pow(10.0, -k); return t; }();
    if (c == end) return 0.f;
    double mant = 0.0;
    int expo = 0, got = 0;
    bool neg = false, expNeg = false, leadingDot = false;
    auto digit = [&](const char* q) { return q != end && *q >= '0' && *q <= '9'; };
    if (*c == '+' || *c == '-') { neg = *c == '-'; c++; if (c != end && *c == '.') leadingDot = true; }
    else if (digit(c)) {}
    else if (*c == '.') leadingDot = true;
    else return 0.f;
    if (!leadingDot) {
        while (digit(c)) { mant *= 10; mant += (int)(*c - '0'); c++; got++; }
        if (got == 0) return 0.f;
    }
    if (c != end && *c == '.') {
        static const double lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
        c++;
        int k = 1;
        while (digit(c)) { mant += (int)(*c - '0') * (k < 8 ? lut[k] : k < 64 ? tenTo[k] : std::pow(10.0, -k)); k++; c++; }
    }
    else if (c != end && *c != 'e' && *c != 'E') c = end;            // trailing text after the integer part: the number so far
    if (c != end && (*c == 'e' || *c == 'E')) {
        c++;
        if (c != end && (*c == '+' || *c == '-')) { expNeg = *c == '-'; c++; }
        else if (!digit(c)) return 0.f;
        got = 0;
        while (digit(c)) { if (expo > 2147483647 / 10) return 0.f; expo = expo * 10 + (int)(*c - '0'); c++; got++; }
        if (expNeg) expo = -expo;
        if (got == 0) return 0.f;
    }
    const double val = (neg ? -1 : 1) * (expo ? std::ldexp(mant * std::pow(5.0, expo), expo) : mant);
    return (float)val;
}

// The subset of Wavefront OBJ the reference gets through tinyobj::LoadObj (triangulate = true): v / vn / vt / f with 1-based and
// negative indices; a quad is split along its shorter diagonal (tiny_obj_loader.h:1429-1524); other statements (o, g, s, usemtl,
// mtllib, comments) do not change the flattened corner order; polygons with more corners are ear-clipped as there.  Every corner needs a normal (the reference indexes attrib.normals unconditionally, scene.cpp:44).
// returns 0, an error, or -1 when the file cannot be opened (the caller skips the object, as scene.cpp:234-240 does)
int load_obj(const std::string& path, Mesh& m) {
    Lines in;
    if (!in.open(path)) return -1;
    std::vector<float> pv, pn, pt;
    struct Corner { int v, t, n; };
    std::vector<Corner> flat, cs;
    pv.reserve(in.size() / 24); pn.reserve(in.size() / 24);
    const auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f' || c == '\r' || c == '\n'; };
    while (in.good()) {
        const char *p, *e;
        in.next_span(p, e);
        // the next whitespace-separated token of the line as [tb, te); false at the end of the line
        const char *tb, *te;
        const auto token = [&]() {
            while (p != e && blank(*p)) p++;
            tb = p;
            while (p != e && !blank(*p)) p++;
            te = p;
            return tb != te;
        };
        if (!token()) continue;
        const size_t klen = (size_t)(te - tb);
        const auto reals = [&](std::vector<float>& dst, int count) {
            for (int k = 0; k < count; k++) dst.push_back(token() ? obj_real(tb, te) : 0.f);
        };
        if (klen == 1 && tb[0] == 'v') reals(pv, 3);
        else if (klen == 2 && tb[0] == 'v' && tb[1] == 'n') reals(pn, 3);
        else if (klen == 2 && tb[0] == 'v' && tb[1] == 't') reals(pt, 2);
        else if (klen == 1 && tb[0] == 'f') {
            cs.clear();
            while (token()) {
                int raw[3] = { 0, 0, 0 };
                bool have[3] = { false, false, false };
                int field = 0;
                const char* q = tb;
                while (q != te && field < 3) {                         // v, v/t, v//n, v/t/n
                    const char* fb = q;
                    while (q != te && *q != '/') q++;
                    if (q != fb) { raw[field] = std::atoi(std::string(fb, q).c_str()); have[field] = true; }
                    field++;
                    if (q != te) q++;
                }
                auto fix = [](bool present, int idx, size_t count) { return !present ? -1 : idx > 0 ? idx - 1 : idx < 0 ? (int)count + idx : -2; };
                Corner c{ fix(have[0], raw[0], pv.size() / 3), fix(have[1], raw[1], pt.size() / 2), fix(have[2], raw[2], pn.size() / 3) };
                if (c.v < 0 || (size_t)c.v >= pv.size() / 3) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("bad vertex index in " + path).c_str());
                if (c.n < 0 || (size_t)c.n >= pn.size() / 3) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("OBJ face corner without a valid normal in " + path).c_str());
                if (c.t < -1 || (c.t >= 0 && (size_t)c.t >= pt.size() / 2)) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("bad texcoord index in " + path).c_str());
                cs.push_back(c);
            }
            if (cs.size() < 3) continue;                              // "Degenerated face": dropped
            if (cs.size() == 3) flat.insert(flat.end(), cs.begin(), cs.end());
            else if (cs.size() == 4) {
                const float* a = &pv[(size_t)cs[0].v * 3]; const float* b = &pv[(size_t)cs[1].v * 3];
                const float* c = &pv[(size_t)cs[2].v * 3]; const float* d = &pv[(size_t)cs[3].v * 3];
                const float e02x = c[0] - a[0], e02y = c[1] - a[1], e02z = c[2] - a[2];
                const float e13x = d[0] - b[0], e13y = d[1] - b[1], e13z = d[2] - b[2];
                const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
                const float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                if (sqr02 < sqr13) for (int q : { 0, 1, 2, 0, 2, 3 }) flat.push_back(cs[q]);
                else for (int q : { 0, 1, 3, 1, 2, 3 }) flat.push_back(cs[q]);
            }
            else {
                // five corners and more: tinyobjloader's ear clipping (tiny_obj_loader.h:1536-1800) in the plane of the two axes
                // picked from the first corner that is not degenerate; the same float expressions, so the same ears in the same
                // order (and, like there, a polygon it cannot finish within its iteration budget yields fewer triangles)
                const size_t np0 = cs.size();
                int ax0 = 1, ax1 = 2;
                for (size_t k = 0; k < np0; k++) {
                    const float* a = &pv[(size_t)cs[k].v * 3]; const float* b = &pv[(size_t)cs[(k + 1) % np0].v * 3]; const float* c = &pv[(size_t)cs[(k + 2) % np0].v * 3];
                    const float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2];
                    const float e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
                    const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
                    const float eps = 1.1920928955078125e-7f;
                    if (cx > eps || cy > eps || cz > eps) {
                        if (!(cx > cy && cx > cz)) { ax0 = 0; if (cz > cx && cz > cy) ax1 = 1; }
                        break;
                    }
                }
                std::vector<Corner> rest(cs);
                size_t guess = 0, budget = np0, previous = np0;
                while (rest.size() > 3 && budget > 0) {
                    const size_t np = rest.size();
                    if (guess >= np) guess -= np;
                    if (previous != np) { previous = np; budget = np; }
                    else budget--;
                    Corner ear[3];
                    float vx[3], vy[3];
                    for (int k = 0; k < 3; k++) {
                        ear[k] = rest[(guess + k) % np];
                        vx[k] = pv[(size_t)ear[k].v * 3 + ax0]; vy[k] = pv[(size_t)ear[k].v * 3 + ax1];
                    }
                    const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
                    const float cross = e0x * e1y - e0y * e1x;
                    const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
                    if (cross * area < 0.f) { guess += 1; continue; }      // "an internal angle"
                    bool overlap = false;
                    for (size_t other = 3; other < np && !overlap; other++) {
                        const Corner& o = rest[(guess + other) % np];
                        const float tx = pv[(size_t)o.v * 3 + ax0], ty = pv[(size_t)o.v * 3 + ax1];
                        bool inside = false;                               // point in the candidate triangle (crossing number)
                        for (int i = 0, j = 2; i < 3; j = i++)
                            if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) inside = !inside;
                        overlap = inside;
                    }
                    if (overlap) { guess += 1; continue; }
                    flat.push_back(ear[0]); flat.push_back(ear[1]); flat.push_back(ear[2]);
                    rest.erase(rest.begin() + (long)((guess + 1) % np));
                }
                if (rest.size() == 3) flat.insert(flat.end(), rest.begin(), rest.end());
            }
        }
    }
    const bool hasTexcoord = !pt.empty();                              // scene.cpp:39,46-49
    m.v.resize(flat.size() * 3); m.n.resize(flat.size() * 3); m.t.resize(flat.size() * 2);
    for (size_t i = 0; i < flat.size(); i++) {
        const Corner& c = flat[i];
        if (hasTexcoord && c.t < 0) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("OBJ face corner without a texcoord in a file that has vt lines: " + path).c_str());
        for (int d = 0; d < 3; d++) { m.v[i * 3 + d] = pv[(size_t)c.v * 3 + d]; m.n[i * 3 + d] = pn[(size_t)c.n * 3 + d]; }
        for (int d = 0; d < 2; d++) m.t[i * 2 + d] = hasTexcoord ? pt[(size_t)c.t * 2 + d] : 0.f;
    }
    return 0;
}

// binary PPM / PGM (P6 / P5, 8 bit) -> linear float RGB, value / 255
int load_ppm(const std::string& path, bool flipRows, std::vector<float>& data, int& w, int& h) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("cannot open image " + path).c_str());
    auto fail = [&](const char* why) { std::fclose(f); return rs_fail(RS_ERR_UNSUPPORTED, (std::string(why) + ": " + path + " (binary PPM P6 / 8 bit, Radiance HDR, PNG, JPEG, BMP and TGA are decoded here; decode other formats in the caller)").c_str()); };
    auto token = [&](std::string& out) {
        out.clear();
        int c = std::fgetc(f);
        for (;;) {
            while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = std::fgetc(f);
            if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(f); continue; }
            break;
        }
        while (c != EOF && c != ' ' && c != '\n' && c != '\r' && c != '\t') { out += (char)c; c = std::fgetc(f); }
        return !out.empty();
    };
    std::string t;
    if (!token(t) || (t != "P6" && t != "P5")) return fail("not a binary PPM / PGM");
    const int comp = t == "P6" ? 3 : 1;                                  // P5: grey, replicated to RGB
    int maxv = 0;
    if (!token(t)) return fail("truncated header"); w = std::atoi(t.c_str());
    if (!token(t)) return fail("truncated header"); h = std::atoi(t.c_str());
    if (!token(t)) return fail("truncated header"); maxv = std::atoi(t.c_str());
    if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 255) return fail("unsupported PPM header");      // the bytes are taken as they are (no rescaling by maxval, as in stb_image)
    std::vector<unsigned char> raw((size_t)w * h * comp);
    if (std::fread(raw.data(), 1, raw.size(), f) != raw.size()) return fail("truncated pixel data");
    std::fclose(f);
    data.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++) {
        const int sy = flipRows ? h - 1 - y : y;
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) data[((size_t)y * w + x) * 3 + c] = (float)raw[((size_t)sy * w + x) * comp + (comp == 3 ? c : 0)] / 255.f;
    }
    return 0;
}

// Radiance picture (.hdr, "32-bit_rle_rgbe", -Y h +X w) -> linear float RGB: mantissa * 2^(e - 136), 0 when e == 0 -- the
// values stbi_loadf returns for such a file (HDR data passes through without the ldr-to-hdr scale).  Scan lines are either all
// flat RGBE quadruples or all run-length coded per channel (marker 2, 2, width as 16 bits; widths 8..32767).
int load_hdr(const std::string& path, bool flipRows, std::vector<float>& data, int& w, int& h) {
    std::string raw;
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("cannot open image " + path).c_str());
        char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) raw.append(buf, got);
        std::fclose(f);
    }
    auto fail = [&](const char* why) { return rs_fail(RS_ERR_UNSUPPORTED, (std::string(why) + ": " + path).c_str()); };
    size_t at = 0;
    auto header_line = [&]() {
        std::string l;
        while (at < raw.size() && raw[at] != '\n') l += raw[at++];
        if (at < raw.size()) at++;
        return l;
    };
    const std::string magic = header_line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") return fail("not a Radiance HDR picture");
    bool rgbe = false;
    for (;;) {
        if (at >= raw.size()) return fail("truncated HDR header");
        const std::string l = header_line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") rgbe = true;
    }
    if (!rgbe) return fail("unsupported HDR pixel format (need 32-bit_rle_rgbe)");
    const std::string res = header_line();
    if (std::sscanf(res.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) return fail("unsupported HDR orientation (need -Y h +X w)");
    const unsigned char* p = reinterpret_cast<const unsigned char*>(raw.data());
    const size_t n = raw.size();
    std::vector<unsigned char> px((size_t)w * h * 4);
    const bool coded = w >= 8 && w < 32768 && at + 4 <= n && p[at] == 2 && p[at + 1] == 2 && !(p[at + 2] & 0x80);
    if (!coded) {
        if (n - at < px.size()) return fail("truncated HDR pixel data");
        std::memcpy(px.data(), p + at, px.size());
    }
    else for (int y = 0; y < h; y++) {
        if (at + 4 > n || p[at] != 2 || p[at + 1] != 2 || ((p[at + 2] << 8) | p[at + 3]) != w) return fail("corrupt HDR scan line");
        at += 4;
        for (int k = 0; k < 4; k++) {
            int x = 0;
            while (x < w) {
                if (at >= n) return fail("truncated HDR pixel data");
                int count = p[at++];
                if (count > 128) {                                    // a run of one value
                    count -= 128;
                    if (count > w - x || at >= n) return fail("corrupt HDR run");
                    const unsigned char v = p[at++];
                    for (int z = 0; z < count; z++) px[((size_t)y * w + x++) * 4 + k] = v;
                }
                else {                                                // literal bytes
                    if (count > w - x || at + count > n) return fail("corrupt HDR run");
                    for (int z = 0; z < count; z++) px[((size_t)y * w + x++) * 4 + k] = p[at++];
                }
            }
        }
    }
    data.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++) {
        const int sy = flipRows ? h - 1 - y : y;
        for (int x = 0; x < w; x++) {
            const unsigned char* q = &px[((size_t)sy * w + x) * 4];
            const float scale = q[3] ? std::ldexp(1.f, (int)q[3] - 136) : 0.f;
            for (int c = 0; c < 3; c++) data[((size_t)y * w + x) * 3 + c] = q[3] ? (float)q[c] * scale : 0.f;
        }
    }
    return 0;
}

// ---- PNG ------------------------------------------------------------------------------------------------------------
// What stbi_loadf(file, .., 3) returns for a PNG (src/image.cpp:17): the 8-bit image -- 16-bit samples keep their high byte,
// 1 / 2 / 4-bit grey is scaled to 0..255, palette entries are expanded, grey is replicated to RGB, alpha and tRNS are dropped
// -- as byte / 255 (ldr-to-hdr gamma 1).  gAMA and other ancillary chunks are ignored and checksums are not verified, as there.
// Decoder: zlib stream of the concatenated IDAT chunks (stored / fixed / dynamic Huffman blocks), the five scan-line filters,
// Adam7 interlacing.

// LSB-first bit reader over the deflate stream
struct BitReader {
    const unsigned char* p; size_t n, at = 0; unsigned acc = 0; int have = 0; bool bad = false;
    int bits(int count) {
        while (have < count) { if (at >= n) { bad = true; return 0; } acc |= (unsigned)p[at++] << have; have += 8; }
        const int v = (int)(acc & ((1u << count) - 1u));
        acc >>= count; have -= count;
        return v;
    }
};
// canonical Huffman code: per length, how many codes and (in order) their symbols
struct Huffman {
    unsigned short count[16], symbol[288];
    bool build(const unsigned char* lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        int left = 1;
        for (int len = 1; len < 16; len++) { left = (left << 1) - count[len]; if (left < 0) return false; }
        unsigned short offs[16]; offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (unsigned short)(offs[len] + count[len]);
        for (int i = 0; i < n; i++) if (lengths[i]) symbol[offs[lengths[i]]++] = (unsigned short)i;
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= br.bits(1);
            if (br.bad) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

bool inflate_zlib(const std::vector<unsigned char>& in, std::vector<unsigned char>& out, size_t expected) {
    if (in.size() < 2 || (in[0] & 15) != 8 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 32)) return false;
    BitReader br{ in.data() + 2, in.size() - 2 };
    static const unsigned short lenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const unsigned char lenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const unsigned short distBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    static const unsigned char distExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    out.clear(); out.reserve(expected);
    for (;;) {
        const int last = br.bits(1), type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {                                              // stored
            br.acc = 0; br.have = 0;
            if (br.at + 4 > br.n) return false;
            const unsigned len = br.p[br.at] | (br.p[br.at + 1] << 8), nlen = br.p[br.at + 2] | (br.p[br.at + 3] << 8);
            br.at += 4;
            if ((len ^ 0xffffu) != nlen || br.at + len > br.n) return false;
            out.insert(out.end(), br.p + br.at, br.p + br.at + len);
            br.at += len;
            if (out.size() > expected) return false;                  // more data than the picture has: not a valid stream
        }
        else if (type == 1 || type == 2) {
            Huffman lit, dist;
            unsigned char lengths[320];
            if (type == 1) {
                for (int i = 0; i < 288; i++) lengths[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; i++) lengths[i] = 5;
                dist.build(lengths, 30);
            }
            else {
                const int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                static const unsigned char order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                unsigned char cl[19] = { 0 };
                for (int i = 0; i < ncode; i++) cl[order[i]] = (unsigned char)br.bits(3);
                Huffman lencode;
                if (!lencode.build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lengths[idx++] = (unsigned char)sym;
                    else {
                        int prev = 0, rep;
                        if (sym == 16) { if (idx == 0) return false; prev = lengths[idx - 1]; rep = 3 + br.bits(2); }
                        else if (sym == 17) rep = 3 + br.bits(3);
                        else rep = 11 + br.bits(7);
                        if (br.bad || idx + rep > nlen + ndist) return false;
                        while (rep--) lengths[idx++] = (unsigned char)prev;
                    }
                }
                if (!lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) { if (out.size() >= expected) return false; out.push_back((unsigned char)sym); }
                else if (sym == 256) break;
                else {
                    if (sym > 285) return false;
                    const int len = lenBase[sym - 257] + br.bits(lenExtra[sym - 257]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds > 29) return false;
                    const size_t d = (size_t)distBase[ds] + (size_t)br.bits(distExtra[ds]);
                    if (br.bad || d > out.size()) return false;
                    const size_t from = out.size() - d;
                    if (out.size() + (size_t)len > expected) return false;
                    for (int i = 0; i < len; i++) out.push_back(out[from + i]);
                }
            }
        }
        else return false;
        if (last) return true;
    }
}

int load_png(const std::string& path, bool flipRows, std::vector<float>& data, int& w, int& h) {
    std::vector<unsigned char> raw;
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("cannot open image " + path).c_str());
        unsigned char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + got);
        std::fclose(f);
    }
    auto fail = [&](const char* why) { return rs_fail(RS_ERR_UNSUPPORTED, (std::string(why) + ": " + path).c_str()); };
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (raw.size() < 8 || std::memcmp(raw.data(), sig, 8) != 0) return fail("not a PNG");
    auto be32 = [&](size_t at) { return ((unsigned)raw[at] << 24) | ((unsigned)raw[at + 1] << 16) | ((unsigned)raw[at + 2] << 8) | raw[at + 3]; };
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, palette;
    bool header = false, ended = false;
    for (size_t at = 8; at + 12 <= raw.size() && !ended;) {
        const size_t len = be32(at);
        const char* tag = reinterpret_cast<const char*>(&raw[at + 4]);
        const size_t body = at + 8;
        if (body + len + 4 > raw.size()) return fail("truncated PNG chunk");
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len != 13) return fail("bad PNG header");
            w = (int)be32(body); h = (int)be32(body + 4);
            depth = raw[body + 8]; ctype = raw[body + 9]; interlace = raw[body + 12];
            if (raw[body + 10] || raw[body + 11] || interlace > 1 || w <= 0 || h <= 0) return fail("bad PNG header");
            header = true;
        }
        else if (!std::memcmp(tag, "PLTE", 4)) palette.assign(raw.begin() + body, raw.begin() + body + len);
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), raw.begin() + body, raw.begin() + body + len);
        else if (!std::memcmp(tag, "IEND", 4)) ended = true;
        else if (!std::memcmp(tag, "CgBI", 4)) return fail("CgBI PNG variants are not decoded");
        at = body + len + 4;
    }
    if (!header) return fail("PNG without a header");
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depthOk = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                       : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!channels || !depthOk) return fail("unsupported PNG colour type / bit depth");
    if (ctype == 3 && palette.size() < 3) return fail("palette PNG without a palette");
    const int bitsPerPixel = channels * depth, bpp = (bitsPerPixel + 7) / 8;
    // passes: the whole image, or the seven Adam7 sub-images
    struct Pass { int x0, y0, dx, dy; };
    static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const Pass whole = { 0, 0, 1, 1 };
    const int numPasses = interlace ? 7 : 1;
    size_t expected = 0;
    for (int pi = 0; pi < numPasses; pi++) {
        const Pass& ps = interlace ? adam7[pi] : whole;
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw > 0 && ph > 0) expected += (size_t)ph * (1 + ((size_t)pw * bitsPerPixel + 7) / 8);
    }
    std::vector<unsigned char> px;
    if (!inflate_zlib(idat, px, expected) || px.size() < expected) return fail("corrupt PNG data stream");
    std::vector<unsigned char> rgb((size_t)w * h * 3);
    const int greyScale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    size_t at = 0;
    std::vector<unsigned char> prevRow, curRow;
    for (int pi = 0; pi < numPasses; pi++) {
        const Pass& ps = interlace ? adam7[pi] : whole;
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw <= 0 || ph <= 0) continue;
        const size_t rowBytes = ((size_t)pw * bitsPerPixel + 7) / 8;
        prevRow.assign(rowBytes, 0);
        for (int y = 0; y < ph; y++) {
            const int filter = px[at++];
            curRow.assign(px.begin() + at, px.begin() + at + rowBytes);
            at += rowBytes;
            if (filter > 4) return fail("bad PNG filter");
            for (size_t i = 0; i < rowBytes; i++) {
                const int a = i >= (size_t)bpp ? curRow[i - bpp] : 0, b = prevRow[i], c = i >= (size_t)bpp ? prevRow[i - bpp] : 0;
                int pred = 0;
                if (filter == 1) pred = a;
                else if (filter == 2) pred = b;
                else if (filter == 3) pred = (a + b) >> 1;
                else if (filter == 4) {
                    const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                }
                curRow[i] = (unsigned char)(curRow[i] + pred);
            }
            const int oy = ps.y0 + y * ps.dy;
            for (int x = 0; x < pw; x++) {
                unsigned char sample[4] = { 0, 0, 0, 0 };
                for (int ch = 0; ch < channels; ch++) {
                    if (depth == 8) sample[ch] = curRow[(size_t)x * channels + ch];
                    else if (depth == 16) sample[ch] = curRow[((size_t)x * channels + ch) * 2];       // the high byte
                    else {
                        const size_t bit = (size_t)x * depth;
                        sample[ch] = (unsigned char)((curRow[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
                    }
                }
                unsigned char* o = &rgb[((size_t)oy * w + ps.x0 + (size_t)x * ps.dx) * 3];
                if (ctype == 3) {
                    const size_t e = (size_t)sample[0] * 3;
                    if (e + 3 > palette.size()) return fail("PNG palette index out of range");
                    o[0] = palette[e]; o[1] = palette[e + 1]; o[2] = palette[e + 2];
                }
                else if (channels <= 2) { const unsigned char g = (unsigned char)(sample[0] * (depth < 8 ? greyScale : 1)); o[0] = o[1] = o[2] = g; }
                else { o[0] = sample[0]; o[1] = sample[1]; o[2] = sample[2]; }
            }
            prevRow.swap(curRow);
        }
    }
    data.resize(rgb.size());
    for (int y = 0; y < h; y++) {
        const int sy = flipRows ? h - 1 - y : y;
        for (int i = 0; i < w * 3; i++) data[((size_t)y * w) * 3 + i] = (float)rgb[((size_t)sy * w) * 3 + i] / 255.f;
    }
    return 0;
}

// ---- JPEG (baseline / extended sequential / progressive, Huffman, 8 bit) -------------------------------------------------------------------
// What stbi_loadf(file, .., 3) returns for such a file: entropy decoding and dequantisation as the standard defines them, the
// integer inverse DCT stb_image uses (jidctint's "slow" form with its rounding: columns keep two extra bits, rows add 65536 +
// (128 << 17) before >> 17), its chroma up-sampling (nearest for 1x, (3 near + far + 2) >> 2 for 2x in one direction, the
// (3 (3 a + b) + (3 c + d) + 8) >> 4 tent for 2x2, replication otherwise) and its fixed-point YCbCr -> RGB
// (external/include/stb_image.h:2267-2340, 3226-3460, 3636-3790); grey files are replicated to RGB, "RGB"-tagged and Adobe
// transform-0 files are taken as RGB; then byte / 255.  Progressive files accumulate their coefficients over the scans (spectral
// selection and successive approximation, ITU T.81 annex G) and are transformed at the end.  Arithmetic-coded, 12-bit and
// four-component (CMYK / YCCK) files are not decoded.
struct JpegHuff { unsigned char bits[17]; unsigned char vals[256]; int mincode[18], maxcode[18], valptr[18]; bool present = false; };

void jpeg_build(JpegHuff& h) {
    int code = 0, k = 0;
    for (int len = 1; len <= 16; len++) {
        h.valptr[len] = k;
        h.mincode[len] = code;
        code += h.bits[len];
        k += h.bits[len];
        h.maxcode[len] = h.bits[len] ? code - 1 : -1;
        code <<= 1;
    }
    h.present = true;
}

struct JpegBits {
    const unsigned char* p; size_t n, at; unsigned acc = 0; int have = 0; int marker = -1;
    int bit() {
        if (!have) {
            unsigned b = 0;
            if (marker < 0 && at < n) {
                b = p[at++];
                if (b == 0xff) {
                    unsigned m = at < n ? p[at] : 0;
                    while (m == 0xff && at + 1 < n) m = p[++at];          // fill bytes
                    if (m == 0) at++;                                     // stuffed zero: a data byte 0xff
                    else { marker = (int)m; at++; b = 0; }                // a marker ends the data: zeros from here on
                }
            }
            acc = b; have = 8;
        }
        have--;
        return (int)((acc >> have) & 1u);
    }
    int receive(int count) { int v = 0; for (int i = 0; i < count; i++) v = (v << 1) | bit(); return v; }
    int extend(int count) {                                               // the signed value of `count` magnitude bits
        if (!count) return 0;
        const int v = receive(count);
        return v < (1 << (count - 1)) ? v - (1 << count) + 1 : v;
    }
    int decode(const JpegHuff& h) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
    void reset() { acc = 0; have = 0; marker = -1; }
};

inline unsigned char jpeg_clamp(int x) { return (unsigned char)(x < 0 ? 0 : x > 255 ? 255 : x); }

// One 8-point pass of the Loeffler-Ligtenberg-Moschytz inverse DCT in 12-bit fixed point (the "slow integer" form of the IJG
// code that stb_image follows; multipliers are round(c * 4096)).  The pass is returned as its even half (four sums / differences of
// the even-indexed inputs) and its odd half (the four rotated odd-indexed inputs); output k is even[k] + odd[3 - k] for k < 4 and
// even[7 - k] - odd[k - 4] above.  The integer expressions and their order are the decoder's, so the pixels are.
struct IdctPass { int even[4], odd[4]; };

inline IdctPass idct_pass(int c0, int c1, int c2, int c3, int c4, int c5, int c6, int c7) {
    IdctPass r;
    const int rot = (c2 + c6) * 2217;                                      // 0.5411961
    const int lowEven = rot + c6 * -7567, highEven = rot + c2 * 3135;      // -1.847759065, 0.765366865
    const int sum = (c0 + c4) * 4096, diff = (c0 - c4) * 4096;
    r.even[0] = sum + highEven; r.even[3] = sum - highEven; r.even[1] = diff + lowEven; r.even[2] = diff - lowEven;
    int o7 = c7, o5 = c5, o3 = c3, o1 = c1;
    const int s73 = o7 + o3, s51 = o5 + o1, s71 = o7 + o1, s53 = o5 + o3;
    const int all = (s73 + s51) * 4816;                                    // 1.175875602
    o7 *= 1223; o5 *= 8410; o3 *= 12586; o1 *= 6149;                       // 0.298631336, 2.053119869, 3.072711026, 1.501321110
    const int t71 = all + s71 * -3685, t53 = all + s53 * -10497;           // -0.899976223, -2.562915447
    const int t73 = s73 * -8034, t51 = s51 * -1597;                        // -1.961570560, -0.390180644
    r.odd[3] = o1 + (t71 + t51); r.odd[2] = o3 + (t53 + t73); r.odd[1] = o5 + (t53 + t51); r.odd[0] = o7 + (t71 + t73);
    return r;
}

void jpeg_idct(unsigned char* out, int stride, const short d[64]) {
    int mid[64];                                                           // after the column pass, two fraction bits kept
    for (int col = 0; col < 8; col++) {
        const short* c = d + col;
        int* m = mid + col;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            const int flat = c[0] * 4;                                     // only the DC term: the column is constant
            for (int k = 0; k < 8; k++) m[8 * k] = flat;
            continue;
        }
        const IdctPass p = idct_pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
        for (int k = 0; k < 4; k++) {
            m[8 * k] = (p.even[k] + 512 + p.odd[3 - k]) >> 10;
            m[8 * (7 - k)] = (p.even[k] + 512 - p.odd[3 - k]) >> 10;
        }
    }
    for (int row = 0; row < 8; row++) {
        const int* m = mid + row * 8;
        unsigned char* o = out + (size_t)row * stride;
        const IdctPass p = idct_pass(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        const int bias = 65536 + (128 << 17);                              // rounding of the 17 fraction bits + the level shift of 128
        for (int k = 0; k < 4; k++) {
            o[k] = jpeg_clamp((p.even[k] + bias + p.odd[3 - k]) >> 17);
            o[7 - k] = jpeg_clamp((p.even[k] + bias - p.odd[3 - k]) >> 17);
        }
    }
}

int load_jpeg(const std::string& path, bool flipRows, std::vector<float>& data, int& w, int& h) {
    std::vector<unsigned char> raw;
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("cannot open image " + path).c_str());
        unsigned char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + got);
        std::fclose(f);
    }
    auto fail = [&](const char* why) { return rs_fail(RS_ERR_UNSUPPORTED, (std::string(why) + ": " + path).c_str()); };
    static const unsigned char zigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
    struct Comp { int id, h, v, tq, td = 0, ta = 0, x, y, w2, h2, pred = 0; std::vector<unsigned char> px; std::vector<short> coef; };
    Comp comp[3];
    int ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0, adobe = -1, rgbTags = 0;
    bool jfif = false, frame = false, done = false, progressive = false;
    unsigned short quant[4][64] = {};
    JpegHuff dc[4], ac[4];
    const size_t n = raw.size();
    if (n < 4 || raw[0] != 0xff || raw[1] != 0xd8) return fail("not a JPEG");
    size_t at = 2;
    int pendingMarker = -1;
    while (!done) {
        int m = pendingMarker;
        pendingMarker = -1;
        if (m < 0) {
            while (at < n && raw[at] != 0xff) at++;                       // (stb skips stray bytes between segments)
            while (at < n && raw[at] == 0xff) at++;
            if (at >= n) return fail("truncated JPEG");
            m = raw[at++];
        }
        if (m == 0xd9) break;                                              // EOI
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;              // markers without a segment
        if (at + 2 > n) return fail("truncated JPEG");
        const size_t len = ((size_t)raw[at] << 8) | raw[at + 1];
        if (len < 2 || at + len > n) return fail("corrupt JPEG segment");
        const unsigned char* q = &raw[at + 2];
        const size_t body = len - 2;
        at += len;
        if (m == 0xc0 || m == 0xc1 || m == 0xc2) {                         // SOF0 / SOF1 / SOF2 (progressive)
            progressive = m == 0xc2;
            if (body < 6 || q[0] != 8) return fail("only 8-bit JPEG files are decoded");
            h = (q[1] << 8) | q[2]; w = (q[3] << 8) | q[4]; ncomp = q[5];
            if (w <= 0 || h <= 0) return fail("bad JPEG size");
            if (ncomp != 1 && ncomp != 3) return fail("only grey and three-component JPEG files are decoded");
            if (body < 6 + 3 * (size_t)ncomp) return fail("corrupt JPEG frame header");
            static const unsigned char tags[3] = { 'R', 'G', 'B' };
            for (int i = 0; i < ncomp; i++) {
                Comp& c = comp[i];
                c.id = q[6 + 3 * i]; c.h = q[7 + 3 * i] >> 4; c.v = q[7 + 3 * i] & 15; c.tq = q[8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("corrupt JPEG frame header");
                if (ncomp == 3 && c.id == tags[i]) rgbTags++;
                hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
            }
            for (int i = 0; i < ncomp; i++) if (hmax % comp[i].h || vmax % comp[i].v) return fail("unsupported JPEG sampling factors");
            mcux = (w + hmax * 8 - 1) / (hmax * 8); mcuy = (h + vmax * 8 - 1) / (vmax * 8);
            for (int i = 0; i < ncomp; i++) {
                Comp& c = comp[i];
                c.x = (w * c.h + hmax - 1) / hmax; c.y = (h * c.v + vmax - 1) / vmax;
                c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8;
                c.px.assign((size_t)c.w2 * c.h2, 0);
                if (progressive) c.coef.assign((size_t)c.w2 * c.h2, 0);      // 64 coefficients per 8x8 block, kept across the scans
            }
            frame = true;
        }
        else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc && m != 0xc4)) return fail("unsupported JPEG coding process");
        else if (m == 0xc4) {                                              // DHT
            size_t o = 0;
            while (o + 17 <= body) {
                const int tc = q[o] >> 4, th = q[o] & 15;
                if (tc > 1 || th > 3) return fail("corrupt JPEG Huffman table");
                JpegHuff& t = tc ? ac[th] : dc[th];
                int total = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; i++) { t.bits[i] = q[o + i]; total += t.bits[i]; }
                if (total > 256 || o + 17 + total > body) return fail("corrupt JPEG Huffman table");
                std::memcpy(t.vals, q + o + 17, (size_t)total);
                jpeg_build(t);
                o += 17 + (size_t)total;
            }
        }
        else if (m == 0xdb) {                                              // DQT
            size_t o = 0;
            while (o < body) {
                const int pq = q[o] >> 4, tq = q[o] & 15;
                if (pq > 1 || tq > 3 || o + 1 + (pq ? 128 : 64) > body) return fail("corrupt JPEG quantisation table");
                for (int i = 0; i < 64; i++) quant[tq][zigzag[i]] = pq ? (unsigned short)((q[o + 1 + 2 * i] << 8) | q[o + 2 + 2 * i]) : q[o + 1 + i];
                o += 1 + (pq ? 128 : 64);
            }
        }
        else if (m == 0xdd) { if (body < 2) return fail("corrupt JPEG restart interval"); restart = (q[0] << 8) | q[1]; }
        else if (m == 0xe0) { if (body >= 5 && !std::memcmp(q, "JFIF\0", 5)) jfif = true; }
        else if (m == 0xee) { if (body >= 12 && !std::memcmp(q, "Adobe\0", 6)) adobe = q[11]; }
        else if (m == 0xda) {                                              // SOS + entropy-coded data
            if (!frame) return fail("JPEG scan before the frame header");
            const int ns = body ? q[0] : 0;
            if (ns < 1 || ns > ncomp || body < 1 + 2 * (size_t)ns + 3) return fail("corrupt JPEG scan header");
            const int specStart = q[1 + 2 * ns], specEnd = q[2 + 2 * ns], succHigh = q[3 + 2 * ns] >> 4, succLow = q[3 + 2 * ns] & 15;
            if (progressive) {
                if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) return fail("corrupt JPEG scan header");
                if ((specStart == 0) != (specEnd == 0) || (specStart && ns != 1)) return fail("corrupt JPEG scan header");
            }
            else if (specStart != 0 || succHigh != 0 || succLow != 0) return fail("corrupt JPEG scan header");
            const bool dcScan = !progressive || specStart == 0, acScan = !progressive || specStart != 0;
            int order[3];
            for (int i = 0; i < ns; i++) {
                int which = -1;
                for (int k = 0; k < ncomp; k++) if (comp[k].id == q[1 + 2 * i]) which = k;
                if (which < 0) return fail("corrupt JPEG scan header");
                comp[which].td = q[2 + 2 * i] >> 4; comp[which].ta = q[2 + 2 * i] & 15;
                if (comp[which].td > 3 || comp[which].ta > 3) return fail("corrupt JPEG scan header");
                if ((dcScan && !(progressive && succHigh) && !dc[comp[which].td].present) || (acScan && !ac[comp[which].ta].present)) return fail("JPEG scan without its Huffman tables");
                order[i] = which;
            }
            JpegBits br{ raw.data(), n, at };
            for (int k = 0; k < ncomp; k++) comp[k].pred = 0;
            int todo = restart ? restart : 0x7fffffff;
            int eobRun = 0;
            bool bad = false;
            auto block = [&](Comp& c, int bx, int by) {
                if (progressive) {                                          // ITU T.81 annex G: this scan's share of the block's coefficients
                    short* d = &c.coef[((size_t)by * (c.w2 >> 3) + bx) * 64];
                    if (dcScan) {
                        if (succHigh == 0) {
                            std::memset(d, 0, 64 * sizeof(short));
                            const int t = br.decode(dc[c.td]);
                            if (t < 0 || t > 15) return false;
                            c.pred += br.extend(t);
                            d[0] = (short)(c.pred * (1 << succLow));
                        }
                        else if (br.bit()) d[0] += (short)(1 << succLow);
                        return true;
                    }
                    if (succHigh == 0) {
                        if (eobRun) { eobRun--; return true; }
                        int k = specStart;
                        do {
                            const int rs = br.decode(ac[c.ta]);
                            if (rs < 0) return false;
                            const int sz = rs & 15, run = rs >> 4;
                            if (sz == 0) {
                                if (run < 15) { eobRun = (1 << run); if (run) eobRun += br.receive(run); eobRun--; break; }
                                k += 16;
                            }
                            else {
                                k += run;
                                if (k > 63) return false;
                                d[zigzag[k++]] = (short)(br.extend(sz) * (1 << succLow));
                            }
                        } while (k <= specEnd);
                        return true;
                    }
                    const short bit = (short)(1 << succLow);                  // refinement of coefficients that are already non-zero
                    auto refine = [&](short* p) {
                        if (br.bit() && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit);
                    };
                    if (eobRun) {
                        eobRun--;
                        for (int k = specStart; k <= specEnd; k++) { short* p = &d[zigzag[k]]; if (*p != 0) refine(p); }
                        return true;
                    }
                    int k = specStart;
                    do {
                        const int rs = br.decode(ac[c.ta]);
                        if (rs < 0) return false;
                        int sz = rs & 15, run = rs >> 4;
                        if (sz == 0) {
                            if (run < 15) { eobRun = (1 << run) - 1; if (run) eobRun += br.receive(run); run = 64; }
                        }
                        else {
                            if (sz != 1) return false;
                            sz = br.bit() ? bit : -bit;
                        }
                        while (k <= specEnd) {
                            short* p = &d[zigzag[k++]];
                            if (*p != 0) refine(p);
                            else {
                                if (run == 0) { *p = (short)sz; break; }
                                run--;
                            }
                        }
                    } while (k <= specEnd);
                    return true;
                }
                short coef[64] = { 0 };
                const unsigned short* dq = quant[c.tq];
                const int t = br.decode(dc[c.td]);
                if (t < 0 || t > 15) return false;
                c.pred += br.extend(t);
                coef[0] = (short)(c.pred * dq[0]);
                for (int k = 1; k < 64;) {
                    const int rs = br.decode(ac[c.ta]);
                    if (rs < 0) return false;
                    const int sz = rs & 15, run = rs >> 4;
                    if (sz == 0) { if (rs != 0xf0) break; k += 16; }
                    else {
                        k += run;
                        if (k > 63) return false;
                        const int zz = zigzag[k++];
                        coef[zz] = (short)(br.extend(sz) * dq[zz]);
                    }
                }
                jpeg_idct(&c.px[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, coef);
                return true;
            };
            auto restart_point = [&]() {
                if (--todo > 0) return true;
                // a restart marker ends the interval: drop the padding bits, expect RSTn, start over
                br.have = 0;
                if (br.marker < 0) {
                    while (br.at + 1 < n && !(raw[br.at] == 0xff && raw[br.at + 1] != 0 && raw[br.at + 1] != 0xff)) br.at++;
                    if (br.at + 1 < n) { br.marker = raw[br.at + 1]; br.at += 2; }
                }
                if (br.marker < 0xd0 || br.marker > 0xd7) return false;    // some other marker: the scan is over
                br.reset();
                for (int k = 0; k < ncomp; k++) comp[k].pred = 0;
                eobRun = 0;
                todo = restart ? restart : 0x7fffffff;
                return true;
            };
            bool ok = true, more = true;
            if (ns == 1) {                                                 // one component: its blocks in raster order
                Comp& c = comp[order[0]];
                const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
                for (int by = 0; by < bh && ok && more; by++)
                    for (int bx = 0; bx < bw && ok && more; bx++) { ok = block(c, bx, by); if (ok) more = restart_point(); }
            }
            else {
                for (int my = 0; my < mcuy && ok && more; my++)
                    for (int mx = 0; mx < mcux && ok && more; mx++) {
                        for (int i = 0; i < ns && ok; i++) {
                            Comp& c = comp[order[i]];
                            for (int y = 0; y < c.v && ok; y++)
                                for (int x = 0; x < c.h && ok; x++) ok = block(c, mx * c.h + x, my * c.v + y);
                        }
                        if (ok) more = restart_point();
                    }
            }
            (void)bad;
            if (!ok) return fail("corrupt JPEG entropy-coded data");
            at = br.at;
            pendingMarker = br.marker;
            if (pendingMarker >= 0xd0 && pendingMarker <= 0xd7) pendingMarker = -1;
        }
        // every other segment (APPn, COM, ...) is skipped
    }
    if (!frame) return fail("JPEG without a frame header");
    if (progressive)                                                       // all scans are in: dequantise and transform every block
        for (int k = 0; k < ncomp; k++) {
            Comp& c = comp[k];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    short* d = &c.coef[((size_t)by * (c.w2 >> 3) + bx) * 64];
                    for (int i = 0; i < 64; i++) d[i] = (short)(d[i] * quant[c.tq][i]);
                    jpeg_idct(&c.px[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, d);
                }
        }
    // up-sample row by row and convert
    const bool isRgb = ncomp == 3 && (rgbTags == 3 || (adobe == 0 && !jfif));
    std::vector<unsigned char> rgb((size_t)w * h * 3), line[3];
    struct Up { int hs, vs, ystep, ypos, wl; const unsigned char* l0; const unsigned char* l1; } up[3];
    for (int k = 0; k < ncomp; k++) {
        up[k].hs = hmax / comp[k].h; up[k].vs = vmax / comp[k].v; up[k].ystep = up[k].vs >> 1; up[k].ypos = 0;
        up[k].wl = (w + up[k].hs - 1) / up[k].hs; up[k].l0 = up[k].l1 = comp[k].px.data();
        line[k].assign((size_t)w + 8, 0);
    }
    for (int j = 0; j < h; j++) {
        const unsigned char* src[3] = { nullptr, nullptr, nullptr };
        for (int k = 0; k < ncomp; k++) {
            Up& u = up[k];
            const bool bot = u.ystep >= (u.vs >> 1);
            const unsigned char* nr = bot ? u.l1 : u.l0;
            const unsigned char* fr = bot ? u.l0 : u.l1;
            unsigned char* o = line[k].data();
            const int wl = u.wl;
            if (u.hs == 1 && u.vs == 1) src[k] = nr;
            else if (u.hs == 1 && u.vs == 2) { for (int i = 0; i < wl; i++) o[i] = (unsigned char)((3 * nr[i] + fr[i] + 2) >> 2); src[k] = o; }
            else if (u.hs == 2 && u.vs == 1) {
                if (wl == 1) o[0] = o[1] = nr[0];
                else {
                    o[0] = nr[0]; o[1] = (unsigned char)((nr[0] * 3 + nr[1] + 2) >> 2);
                    int i = 1;
                    for (; i < wl - 1; i++) { const int t = 3 * nr[i] + 2; o[i * 2] = (unsigned char)((t + nr[i - 1]) >> 2); o[i * 2 + 1] = (unsigned char)((t + nr[i + 1]) >> 2); }
                    o[i * 2] = (unsigned char)((nr[wl - 2] * 3 + nr[wl - 1] + 2) >> 2); o[i * 2 + 1] = nr[wl - 1];
                }
                src[k] = o;
            }
            else if (u.hs == 2 && u.vs == 2) {
                if (wl == 1) o[0] = o[1] = (unsigned char)((3 * nr[0] + fr[0] + 2) >> 2);
                else {
                    int t1 = 3 * nr[0] + fr[0];
                    o[0] = (unsigned char)((t1 + 2) >> 2);
                    for (int i = 1; i < wl; i++) {
                        const int t0 = t1;
                        t1 = 3 * nr[i] + fr[i];
                        o[i * 2 - 1] = (unsigned char)((3 * t0 + t1 + 8) >> 4);
                        o[i * 2] = (unsigned char)((3 * t1 + t0 + 8) >> 4);
                    }
                    o[wl * 2 - 1] = (unsigned char)((t1 + 2) >> 2);
                }
                src[k] = o;
            }
            else { for (int i = 0; i < wl; i++) for (int r = 0; r < u.hs; r++) if (i * u.hs + r < w + 8) o[i * u.hs + r] = nr[i]; src[k] = o; }
            if (++u.ystep >= u.vs) {
                u.ystep = 0;
                u.l0 = u.l1;
                if (++u.ypos < comp[k].y) u.l1 += comp[k].w2;
            }
        }
        unsigned char* o = &rgb[(size_t)j * w * 3];
        if (ncomp == 1) for (int i = 0; i < w; i++) o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = src[0][i];
        else if (isRgb) for (int i = 0; i < w; i++) { o[3 * i] = src[0][i]; o[3 * i + 1] = src[1][i]; o[3 * i + 2] = src[2][i]; }
        else for (int i = 0; i < w; i++) {                                 // fixed point: constants (int)(x * 4096 + 0.5) << 8
            const int yf = (src[0][i] << 20) + (1 << 19), cb = src[1][i] - 128, cr = src[2][i] - 128;
            int r = yf + cr * (5743 << 8);
            int g = yf + cr * -(2925 << 8) + ((cb * -(1410 << 8)) & (int)0xffff0000);
            int b = yf + cb * (7258 << 8);
            r >>= 20; g >>= 20; b >>= 20;
            o[3 * i] = jpeg_clamp(r); o[3 * i + 1] = jpeg_clamp(g); o[3 * i + 2] = jpeg_clamp(b);
        }
    }
    data.resize(rgb.size());
    for (int y = 0; y < h; y++) {
        const int sy = flipRows ? h - 1 - y : y;
        for (int i = 0; i < w * 3; i++) data[((size_t)y * w) * 3 + i] = (float)rgb[((size_t)sy * w) * 3 + i] / 255.f;
    }
    return 0;
}

// ---- TGA ------------------------------------------------------------------------------------------------------------
// Truevision TGA as stb_image reads it (external/include/stb_image.h:5520-5720): image types 1 / 2 / 3 (colour-mapped, true
// colour, grey) and their run-length coded forms 9 / 10 / 11; 8, 15 / 16 (5-5-5, scaled v * 255 / 31), 24 and 32 bits; colour
// maps of those depths; rows bottom-up unless bit 5 of the descriptor says top-down; BGR order in the file.  To RGB as for the
// other formats (grey replicated, alpha dropped), byte / 255.  The format has no signature: files are taken as TGA by name.
int load_tga(const std::string& path, bool flipRows, std::vector<float>& data, int& w, int& h) {
    std::vector<unsigned char> raw;
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("cannot open image " + path).c_str());
        unsigned char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + got);
        std::fclose(f);
    }
    auto fail = [&](const char* why) { return rs_fail(RS_ERR_UNSUPPORTED, (std::string(why) + ": " + path).c_str()); };
    if (raw.size() < 18) return fail("truncated TGA header");
    const int idLen = raw[0], indexed = raw[1];
    int type = raw[2];
    const int palStart = raw[3] | (raw[4] << 8), palLen = raw[5] | (raw[6] << 8), palBits = raw[7];
    w = raw[12] | (raw[13] << 8); h = raw[14] | (raw[15] << 8);
    const int bpp = raw[16], bottomUp = 1 - ((raw[17] >> 5) & 1);
    const bool rle = type >= 8;
    if (rle) type -= 8;
    if (type < 1 || type > 3 || w <= 0 || h <= 0 || (indexed != 0) != (type == 1)) return fail("unsupported TGA image type");
    auto channels = [](int bits, bool grey, bool& rgb16) {
        rgb16 = false;
        switch (bits) {
        case 8: return 1;
        case 16: if (grey) return 2;  // grey + alpha
                 // fall through
        case 15: rgb16 = true; return 3;
        case 24: return 3;
        case 32: return 4;
        default: return 0;
        }
    };
    bool rgb16 = false;
    const int comp = indexed ? channels(palBits, false, rgb16) : channels(bpp, type == 3, rgb16);
    if (!comp || (indexed && bpp != 8 && bpp != 16)) return fail("unsupported TGA pixel format");
    size_t at = 18 + (size_t)idLen;
    auto need = [&](size_t count) { return at + count <= raw.size(); };
    auto rgb555 = [&](unsigned char* o) {
        const unsigned px = raw[at] | (raw[at + 1] << 8);
        at += 2;
        o[0] = (unsigned char)((((px >> 10) & 31) * 255) / 31); o[1] = (unsigned char)((((px >> 5) & 31) * 255) / 31); o[2] = (unsigned char)(((px & 31) * 255) / 31);
    };
    std::vector<unsigned char> palette;
    if (indexed) {
        at += (size_t)palStart;                                            // (stb skips this many bytes before the colour map)
        palette.resize((size_t)palLen * comp);
        if (rgb16) { if (!need((size_t)palLen * 2)) return fail("truncated TGA colour map"); for (int i = 0; i < palLen; i++) rgb555(&palette[(size_t)i * 3]); }
        else { if (!need(palette.size())) return fail("truncated TGA colour map"); std::memcpy(palette.data(), &raw[at], palette.size()); at += palette.size(); }
    }
    std::vector<unsigned char> px((size_t)w * h * comp);
    unsigned char cur[4] = { 0, 0, 0, 0 };
    int count = 0;
    bool repeating = false, read = true;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        if (rle) {
            if (count == 0) {
                if (!need(1)) return fail("truncated TGA pixel data");
                const int cmd = raw[at++];
                count = 1 + (cmd & 127); repeating = (cmd >> 7) != 0; read = true;
            }
            else if (!repeating) read = true;
        }
        else read = true;
        if (read) {
            if (indexed) {
                if (!need(bpp == 8 ? 1 : 2)) return fail("truncated TGA pixel data");
                int idx = bpp == 8 ? raw[at] : (raw[at] | (raw[at + 1] << 8));
                at += bpp == 8 ? 1 : 2;
                if (idx >= palLen) idx = 0;
                for (int j = 0; j < comp; j++) cur[j] = palLen ? palette[(size_t)idx * comp + j] : 0;
            }
            else if (rgb16) { if (!need(2)) return fail("truncated TGA pixel data"); rgb555(cur); }
            else { if (!need((size_t)comp)) return fail("truncated TGA pixel data"); for (int j = 0; j < comp; j++) cur[j] = raw[at++]; }
            read = false;
        }
        for (int j = 0; j < comp; j++) px[i * comp + j] = cur[j];
        count--;
    }
    data.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++) {
        int sy = bottomUp ? h - 1 - y : y;                                 // file order -> top-down ...
        if (flipRows) sy = bottomUp ? y : h - 1 - y;                       // ... and the loader's vertical flip on top of it
        for (int x = 0; x < w; x++) {
            const unsigned char* q = &px[((size_t)sy * w + x) * comp];
            unsigned char r, g, b;
            if (comp <= 2) r = g = b = q[0];
            else if (rgb16) { r = q[0]; g = q[1]; b = q[2]; }
            else { r = q[2]; g = q[1]; b = q[0]; }
            float* o = &data[((size_t)y * w + x) * 3];
            o[0] = (float)r / 255.f; o[1] = (float)g / 255.f; o[2] = (float)b / 255.f;
        }
    }
    return 0;
}

// ---- BMP ------------------------------------------------------------------------------------------------------------
// Windows / OS2 bitmaps as the reference's stb_image v2.21 reads them (external/include/stb_image.h:5122-5400): header sizes 12,
// 40, 56, 108 and 124; 1 / 4 / 8-bit palette pictures, 16-bit (5-5-5 or bit fields), 24-bit and 32-bit (bit fields or 8-8-8-8);
// no run-length coding; bottom-up unless the height is negative.  Channels of other widths than 8 bits are expanded by bit
// replication as there.  Its reading position is followed literally -- bytes past the end of the file read as zero, and the
// pixel offset is applied relative to where the header parser stopped, which for a 40-byte header followed by three mask words
// skips twelve bytes of pixel data, as that version does.
int load_bmp(const std::string& path, bool flipRows, std::vector<float>& data, int& w, int& h) {
    std::vector<unsigned char> raw;
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("cannot open image " + path).c_str());
        unsigned char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + got);
        std::fclose(f);
    }
    auto fail = [&](const char* why) { return rs_fail(RS_ERR_UNSUPPORTED, (std::string(why) + ": " + path).c_str()); };
    size_t at = 0;
    auto get8 = [&]() -> unsigned { return at < raw.size() ? raw[at++] : 0u; };
    auto get16 = [&]() -> unsigned { const unsigned a = get8(); return a | (get8() << 8); };
    auto get32 = [&]() -> unsigned { const unsigned a = get16(); return a | (get16() << 16); };
    auto skip = [&](long long count) { if (count < 0) at = raw.size(); else at = (size_t)std::min<unsigned long long>(raw.size(), at + (unsigned long long)count); };
    if (get8() != 'B' || get8() != 'M') return fail("not a BMP");
    get32(); get16(); get16();
    const long long offset = (int)get32();
    const int hsz = (int)get32();
    unsigned mr = 0, mg = 0, mb = 0, ma = 0;
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return fail("unknown BMP header");
    int height;
    if (hsz == 12) { w = (int)get16(); height = (int)get16(); }
    else { w = (int)get32(); height = (int)get32(); }
    if (get16() != 1) return fail("bad BMP");
    const int bpp = (int)get16();
    if (hsz != 12) {
        const int compress = (int)get32();
        if (compress == 1 || compress == 2) return fail("run-length coded BMP files are not decoded");
        get32(); get32(); get32(); get32(); get32();
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) { get32(); get32(); get32(); get32(); }
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) {
                    if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; }
                    else { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
                }
                else if (compress == 3) {
                    mr = get32(); mg = get32(); mb = get32();
                    if (mr == mg && mg == mb) return fail("bad BMP masks");
                }
                else return fail("bad BMP compression");
            }
        }
        else {
            mr = get32(); mg = get32(); mb = get32(); ma = get32();
            get32();
            for (int i = 0; i < 12; i++) get32();
            if (hsz == 124) { get32(); get32(); get32(); get32(); }
        }
    }
    const bool bottomUp = height > 0;
    h = std::abs(height);
    if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) return fail("bad BMP size");
    int psize = 0;
    if (hsz == 12) { if (bpp < 24) psize = (int)((offset - 14 - 24) / 3); }
    else if (bpp < 16) psize = (int)((offset - 14 - hsz) >> 2);
    std::vector<unsigned char> rgb((size_t)w * h * 3);
    size_t z = 0;
    if (bpp < 16) {
        if (psize <= 0 || psize > 256) return fail("corrupt BMP palette");
        unsigned char pal[256][3];
        for (int i = 0; i < psize; i++) { pal[i][2] = (unsigned char)get8(); pal[i][1] = (unsigned char)get8(); pal[i][0] = (unsigned char)get8(); if (hsz != 12) get8(); }
        for (int i = psize; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = 0;       // (uninitialised there)
        skip(offset - 14 - hsz - (long long)psize * (hsz == 12 ? 3 : 4));
        int width;
        if (bpp == 1) width = (w + 7) >> 3;
        else if (bpp == 4) width = (w + 1) >> 1;
        else if (bpp == 8) width = w;
        else return fail("bad BMP bit depth");
        const int pad = (-width) & 3;
        for (int j = 0; j < h; j++) {
            if (bpp == 1) {
                int bit = 7;
                unsigned v = get8();
                for (int i = 0; i < w; i++) {
                    const int c = (int)((v >> bit) & 1u);
                    rgb[z++] = pal[c][0]; rgb[z++] = pal[c][1]; rgb[z++] = pal[c][2];
                    if (i + 1 == w) break;
                    if (--bit < 0) { bit = 7; v = get8(); }
                }
            }
            else for (int i = 0; i < w; i += 2) {
                unsigned v = get8(), v2 = 0;
                if (bpp == 4) { v2 = v & 15u; v >>= 4; }
                rgb[z++] = pal[v][0]; rgb[z++] = pal[v][1]; rgb[z++] = pal[v][2];
                if (i + 1 == w) break;
                v = bpp == 8 ? get8() : v2;
                rgb[z++] = pal[v][0]; rgb[z++] = pal[v][1]; rgb[z++] = pal[v][2];
            }
            skip(pad);
        }
    }
    else {
        skip(offset - 14 - hsz);
        const int width = bpp == 24 ? 3 * w : bpp == 16 ? 2 * w : 0;
        const int pad = (-width) & 3;
        int easy = 0;
        if (bpp == 24) easy = 1;
        else if (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0x00ff0000u && ma == 0xff000000u) easy = 2;
        else if (bpp != 16 && bpp != 32) return fail("bad BMP bit depth");
        auto high_bit = [](unsigned v) { int n = -1; while (v) { n++; v >>= 1; } return n; };
        auto bit_count = [](unsigned v) { int n = 0; while (v) { n += (int)(v & 1u); v >>= 1; } return n; };
        int rshift = 0, gshift = 0, bshift = 0, rcount = 0, gcount = 0, bcount = 0;
        if (!easy) {
            if (!mr || !mg || !mb) return fail("bad BMP masks");
            rshift = high_bit(mr) - 7; rcount = bit_count(mr);
            gshift = high_bit(mg) - 7; gcount = bit_count(mg);
            bshift = high_bit(mb) - 7; bcount = bit_count(mb);
            if (rcount > 8 || gcount > 8 || bcount > 8) return fail("bad BMP masks");
        }
        auto channel = [](unsigned v, int shift, int bits) {               // an n-bit field made 8 bits wide by bit replication
            static const unsigned mul[9] = { 0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01 };
            static const unsigned shr[9] = { 0, 0, 0, 1, 0, 2, 4, 6, 0 };
            if (shift < 0) v <<= -shift; else v >>= shift;
            v >>= (8 - bits);
            return (unsigned char)((v * mul[bits]) >> shr[bits]);
        };
        for (int j = 0; j < h; j++) {
            for (int i = 0; i < w; i++) {
                if (easy) {
                    rgb[z + 2] = (unsigned char)get8(); rgb[z + 1] = (unsigned char)get8(); rgb[z] = (unsigned char)get8();
                    z += 3;
                    if (easy == 2) get8();
                }
                else {
                    const unsigned v = bpp == 16 ? get16() : get32();
                    rgb[z++] = channel(v & mr, rshift, rcount); rgb[z++] = channel(v & mg, gshift, gcount); rgb[z++] = channel(v & mb, bshift, bcount);
                }
            }
            skip(pad);
        }
    }
    data.resize(rgb.size());
    for (int y = 0; y < h; y++) {
        int sy = bottomUp ? h - 1 - y : y;
        if (flipRows) sy = bottomUp ? y : h - 1 - y;
        for (int i = 0; i < w * 3; i++) data[((size_t)y * w) * 3 + i] = (float)rgb[((size_t)sy * w) * 3 + i] / 255.f;
    }
    return 0;
}

// by content, not by file name: "P6" = binary PPM, "#?" = Radiance HDR, 0x89 "PNG" = PNG, 0xff 0xd8 = JPEG, "BM" = BMP; TGA by its file name
int load_image(const std::string& path, bool flipRows, std::vector<float>& data, int& w, int& h) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return rs_fail(RS_ERR_INVALID_ARGUMENT, ("cannot open image " + path).c_str());
    unsigned char m[2] = { 0, 0 };
    const size_t got = std::fread(m, 1, 2, f);
    std::fclose(f);
    if (got == 2 && m[0] == '#' && m[1] == '?') return load_hdr(path, flipRows, data, w, h);
    if (got == 2 && m[0] == 0x89 && m[1] == 'P') return load_png(path, flipRows, data, w, h);
    if (got == 2 && m[0] == 0xff && m[1] == 0xd8) return load_jpeg(path, flipRows, data, w, h);
    if (got == 2 && m[0] == 'B' && m[1] == 'M') return load_bmp(path, flipRows, data, w, h);
    if (path.size() >= 4) {
        std::string ext = path.substr(path.size() - 4);
        for (char& c : ext) c = (char)std::tolower((unsigned char)c);
        if (ext == ".tga") return load_tga(path, flipRows, data, w, h);
    }
    return load_ppm(path, flipRows, data, w, h);
}

}  // namespace

struct rs_scene_file {
    std::vector<float> vertices, normals, texcoords;
    std::vector<int> materialIds;
    std::vector<rs_material> materials;
    std::vector<std::vector<float>> texData;
    std::vector<rs_texture> textures;
    std::map<std::string, int> textureIds, materialMap;
    int envMapTexId = -1;
    rs_camera camera{};
    int iterations = 0, traceDepth = 0;
    std::string imageName;
    std::vector<std::string> skipped;                                  // objects whose mesh file could not be opened
};

namespace {

// The reference opens the names as written, i.e. relative to the working directory; a name that is not there is also tried next to
// the scene file.
std::string resolve(const std::string& dir, const std::string& name) {
    if (name.empty() || name[0] == '/' || dir.empty()) return name;
    if (FILE* f = std::fopen(name.c_str(), "rb")) { std::fclose(f); return name; }
    return dir + name;
}

int add_texture(rs_scene_file* s, const std::string& dir, const std::string& name, bool flip, int* id) {   // Scene::addTexture (:356-369)
    auto it = s->textureIds.find(name);
    if (it != s->textureIds.end()) { *id = it->second; return 0; }
    std::vector<float> data; int w = 0, h = 0;
    RS_TRY(load_image(resolve(dir, name), flip, data, w, h));
    s->texData.push_back(std::move(data));
    *id = (int)s->texData.size() - 1;
    s->textureIds[name] = *id;
    s->textures.push_back(rs_texture{ w, h, nullptr });
    return 0;
}

rs_material default_material() {                                      // src/material.h:258-267
    rs_material m;
    m.type = 0; m.baseColor[0] = m.baseColor[1] = m.baseColor[2] = .9f; m.metallic = 0.f; m.roughness = 1.f; m.ior = 1.5f;
    m.baseColorMapId = m.metallicMapId = m.roughnessMapId = m.normalMapId = -1;
    return m;
}

}  // namespace

extern "C" {

int rs_build_transformation_matrix(const float* translation, const float* rotation, const float* scl, float* out16) {
    if (!translation || !rotation || !scl || !out16) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_build_transformation_matrix: null argument");
    const M4 m = build_transformation_matrix(ld3(translation), ld3(rotation), ld3(scl));
    std::memcpy(out16, &m, sizeof m);
    return 0;
}

int rs_bake_instance(const float* translation, const float* rotation, const float* scl, int n, const float* vertsIn, const float* normalsIn,
                     float* vertsOut, float* normalsOut) {
    if (!translation || !rotation || !scl || n < 0 || (n > 0 && (!vertsIn || !normalsIn || !vertsOut || !normalsOut)))
        return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_bake_instance: bad argument");
    bake(build_transformation_matrix(ld3(translation), ld3(rotation), ld3(scl)), n, vertsIn, normalsIn, vertsOut, normalsOut);
    return 0;
}

int rs_scene_file_free(rs_scene_file* s) { delete s; return 0; }

int rs_scene_file_load(const char* path, rs_scene_file** out) {
    if (!path || !out) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_file_load: null argument");
    *out = nullptr;
    Lines fp;
    if (!fp.open(path)) return rs_fail(RS_ERR_INVALID_ARGUMENT, (std::string("Error reading from file ") + path).c_str());
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);      // file names in the scene are relative to it

    rs_scene_file* s = new rs_scene_file();
    auto bail = [&](int e) { delete s; return e; };
    std::map<std::string, Mesh> meshPool;                              // Resource::meshDataPool
    static const std::map<std::string, int> typeOf = { { "Lambertian", 0 }, { "MetallicWorkflow", 1 }, { "Dielectric", 2 }, { "Light", 4 } };   // :16-21

    try {
    std::string line;
    while (fp.good()) {
        line = fp.next();
        if (line.empty()) continue;
        const std::vector<std::string> tokens = tokenize(line);
        if (tokens.empty()) continue;
        if (tokens[0] == "Material" && tokens.size() >= 2) {                                     // loadMaterial (:371-433): exactly six lines follow
            rs_material material = default_material();
            for (int i = 0; i < 6; i++) {
                line = fp.next();
                const std::vector<std::string> t = tokenize(line);
                if (t.size() < 2) continue;
                if (t[0] == "Type") {
                    auto it = typeOf.find(t[1]);
                    material.type = it == typeOf.end() ? 0 : it->second;   // std::map::operator[] of an unknown token yields 0 = Lambertian (:386)
                }
                else if (t[0] == "BaseColor") {
                    if (t.size() > 2) { if (t.size() < 4) return bail(rs_fail(RS_ERR_INVALID_ARGUMENT, "BaseColor needs three numbers")); for (int k = 0; k < 3; k++) material.baseColor[k] = std::stof(t[1 + k]); }
                    else if (t[1] == "Procedural") material.baseColorMapId = -2;
                    else if (int e = add_texture(s, dir, t[1], true, &material.baseColorMapId)) return bail(e);
                }
                else if (t[0] == "Metallic") {
                    if (std::isdigit((unsigned char)t[1][t[1].length() - 1])) material.metallic = std::stof(t[1]);
                    else if (int e = add_texture(s, dir, t[1], true, &material.metallicMapId)) return bail(e);
                }
                else if (t[0] == "Roughness") {
                    if (std::isdigit((unsigned char)t[1][t[1].length() - 1])) material.roughness = std::stof(t[1]);
                    else if (int e = add_texture(s, dir, t[1], true, &material.roughnessMapId)) return bail(e);
                }
                else if (t[0] == "Ior") material.ior = std::stof(t[1]);
                else if (t[0] == "NormalMap") {
                    if (t[1] != "Null") if (int e = add_texture(s, dir, t[1], true, &material.normalMapId)) return bail(e);
                }
            }
            s->materialMap[tokens[1]] = (int)s->materials.size();
            s->materials.push_back(material);
        }
        else if (tokens[0] == "Object" && tokens.size() >= 2) {                                    // loadModel (:222-283)
            line = fp.next();
            const std::string filename = line;
            auto pool = meshPool.find(filename);
            if (pool == meshPool.end()) {
                if (filename.find(".obj") == std::string::npos) return bail(rs_fail(RS_ERR_UNSUPPORTED, ("only OBJ meshes are read here: " + filename).c_str()));
                Mesh mesh;
                const int e = load_obj(resolve(dir, filename), mesh);
                if (e > 0) return bail(e);
                if (e < 0) {                                          // "[Fail to load, skipped]" (:234-240)
                    while (!line.empty() && fp.good()) line = fp.next();
                    s->skipped.push_back(filename);
                    continue;
                }
                pool = meshPool.emplace(filename, std::move(mesh)).first;
            }
            int materialId = 0;
            line = fp.next();
            if (!line.empty() && fp.good()) {
                const std::vector<std::string> t = tokenize(line);
                if (t.size() < 2) return bail(rs_fail(RS_ERR_INVALID_ARGUMENT, "Object: expected `Material <name>`"));
                if (t[1] == "Null") { materialId = (int)s->materials.size(); s->materials.push_back(default_material()); }
                else {
                    auto it = s->materialMap.find(t[1]);
                    if (it == s->materialMap.end()) return bail(rs_fail(RS_ERR_INVALID_ARGUMENT, ("Material " + t[1] + " doesn't exist").c_str()));
                    materialId = it->second;
                }
            }
            f3 translation = splat(0.f), rotation = splat(0.f), scl = splat(0.f);                   // glm::vec3() members of ModelInstance (Q13)
            line = fp.next();
            while (!line.empty() && fp.good()) {
                const std::vector<std::string> t = tokenize(line);
                if (t.size() >= 4) {
                    const f3 v = mk3(std::stof(t[1]), std::stof(t[2]), std::stof(t[3]));
                    if (t[0] == "Translate") translation = v;
                    else if (t[0] == "Rotate") rotation = v;
                    else if (t[0] == "Scale") scl = v;
                }
                line = fp.next();
            }
            // buildDevData (:161-176): transformed corners appended in instance order, one material id per triangle
            const Mesh& mesh = pool->second;
            const int n = (int)(mesh.v.size() / 3);
            const size_t at = s->vertices.size();
            s->vertices.resize(at + mesh.v.size()); s->normals.resize(at + mesh.n.size());
            bake(build_transformation_matrix(translation, rotation, scl), n, mesh.v.data(), mesh.n.data(), s->vertices.data() + at, s->normals.data() + at);
            s->texcoords.insert(s->texcoords.end(), mesh.t.begin(), mesh.t.end());
            s->materialIds.insert(s->materialIds.end(), (size_t)(n / 3), materialId);
        }
        else if (tokens[0] == "Camera") {                                                        // loadCamera (:285-354)
            float fovy = 0.f;
            rs_camera& cam = s->camera;
            for (int i = 0; i < 8; i++) {
                line = fp.next();
                const std::vector<std::string> t = tokenize(line);
                if (t.size() < 2) continue;
                if (t[0] == "Resolution" && t.size() >= 3) { cam.resolution[0] = std::stoi(t[1]); cam.resolution[1] = std::stoi(t[2]); }
                else if (t[0] == "FovY") fovy = std::stof(t[1]);
                else if (t[0] == "LensRadius") cam.lensRadius = std::stof(t[1]);
                else if (t[0] == "FocalDist") cam.focalDist = std::stof(t[1]);
                else if (t[0] == "Sample") s->iterations = std::stoi(t[1]);
                else if (t[0] == "Depth") s->traceDepth = std::stoi(t[1]);
                else if (t[0] == "File") s->imageName = t[1];
            }
            line = fp.next();
            while (!line.empty() && fp.good()) {
                const std::vector<std::string> t = tokenize(line);
                if (t.size() >= 4) {
                    float* dst = t[0] == "Eye" ? cam.position : t[0] == "Rotation" ? cam.rotation : t[0] == "Up" ? cam.up : nullptr;
                    if (dst) for (int k = 0; k < 3; k++) dst[k] = std::stof(t[1 + k]);
                }
                line = fp.next();
            }
            if (cam.resolution[0] <= 0 || cam.resolution[1] <= 0) return bail(rs_fail(RS_ERR_INVALID_ARGUMENT, "Camera: missing Resolution"));
            const float yscaled = tanf(fovy * (kPi / 180));                                      // :343-348
            const float xscaled = (yscaled * (float)cam.resolution[0]) / (float)cam.resolution[1];
            const float fovx = (atanf(xscaled) * 180) / kPi;
            cam.fov[0] = fovx; cam.fov[1] = fovy;
            cam.tanFovY = tanf(radians(fovy * 0.5f));
            cam.pixelLength[0] = cam.pixelLength[1] = 0.f;
            if (int e = rs_camera_update(&cam)) return bail(e);
        }
        else if (tokens[0] == "EnvMap" && tokens.size() >= 2) {                                   // :122-128: not flipped
            if (tokens[1] != "Null") if (int e = add_texture(s, dir, tokens[1], false, &s->envMapTexId)) return bail(e);
        }
    }
    } catch (const std::bad_alloc&) {
        return bail(rs_fail(RS_ERR_INVALID_ARGUMENT, "out of memory while reading the scene (a picture or mesh of absurd size?)"));
    } catch (const std::exception& e) {                                // std::stof / std::stoi on malformed numbers
        return bail(rs_fail(RS_ERR_INVALID_ARGUMENT, (std::string("malformed number in scene file: ") + e.what()).c_str()));
    }
    if (s->vertices.empty()) return bail(rs_fail(RS_ERR_INVALID_ARGUMENT, "No mesh data loaded"));      // scene.cpp:192-195
    for (size_t i = 0; i < s->textures.size(); i++) s->textures[i].data = s->texData[i].data();
    *out = s;
    return 0;
}

int rs_scene_file_get(const rs_scene_file* s, rs_scene_file_view* v) {
    if (!s || !v) return rs_fail(RS_ERR_INVALID_ARGUMENT, "rs_scene_file_get: null argument");
    v->numPrims = (int)(s->vertices.size() / 9);
    v->vertices = s->vertices.data(); v->normals = s->normals.data(); v->texcoords = s->texcoords.data();
    v->materialIds = s->materialIds.data();
    v->numMaterials = (int)s->materials.size(); v->materials = s->materials.data();
    v->numTextures = (int)s->textures.size(); v->textures = s->textures.data(); v->envMapTexId = s->envMapTexId;
    v->camera = s->camera; v->iterations = s->iterations; v->traceDepth = s->traceDepth; v->imageName = s->imageName.c_str();
    v->numSkippedObjects = (int)s->skipped.size();
    return 0;
}

}  // extern "C"
