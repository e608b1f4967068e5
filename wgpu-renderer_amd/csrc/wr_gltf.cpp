// wr_gltf.cpp -- GLB (binary glTF 2.0) import on the host, reproducing the reference's
// loader with its quirks (src/core/resources.rs:17-136 load_model_gltf,
// src/core/mesh.rs:13-112 Mesh::from_gltf):
//   * meshes come from document.meshes(); node transforms are ignored (resources.rs:31-33);
//     one mesh per primitive, in document order (mesh.rs:20-21);
//   * tangent = TANGENT.xyz * TANGENT.w, bitangent = normal x tangent (mesh.rs:56-64); a
//     primitive without TANGENT gets NaN tangents and bitangents, because calc_tangents runs
//     before the indices are read and divides by a zero count (mesh.rs:69,151-155);
//   * a material's missing texture slot falls back to image 0 (resources.rs:46,74,101);
//   * RGB images are padded to RGBA with alpha 255 (resources.rs:184-191); images with other
//     channel counts or 16-bit channels are rejected (resources.rs:51,138-151);
//   * base colour is sRGB, normal and metallic-roughness are linear (resources.rs:62,89,116);
//   * primitive.material defaults to 0 (mesh.rs:23).
// Images: PNG (8-bit RGB / RGBA / palette, not interlaced) is decoded here with zlib; where
// the reference uses the `image` crate the decoded texels are the same bytes.  JPEG (baseline,
// extended sequential and progressive, three components) goes through wr_jpeg.cpp, whose
// texels are close to, but not pinned to, the reference's decoder.  Pure host code: nothing
// here touches the device.

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/wr_abi.h"

int wr_decode_jpeg(const uint8_t* data, size_t n, uint32_t* w, uint32_t* h, std::vector<uint8_t>& rgba, std::string& err);

namespace {

// ---------------------------------------------------------------- JSON (just enough for glTF)
struct JVal {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;

    const JVal* get(const char* key) const {
        if (type != Obj) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const char* key) const { return get(key) != nullptr; }
    double number(const char* key, double dflt) const {
        const JVal* v = get(key);
        return (v && v->type == Num) ? v->num : dflt;
    }
    long integer(const char* key, long dflt) const {
        const double d = number(key, (double)dflt);
        // anything that is not an integer of moderate size reads as -1, which every caller rejects
        if (!(d >= -1.0 && d <= 9.0e15) || d != (double)(long)d) return -1;
        return (long)d;
    }
    // A size or offset: a non-negative integer below 2^53, else `ok` turns false.
    size_t unsigned_integer(const char* key, size_t dflt, bool& ok) const {
        const JVal* v = get(key);
        if (!v) return dflt;
        if (v->type != Num || !(v->num >= 0.0 && v->num <= 9.0e15) || v->num != (double)(unsigned long long)v->num) { ok = false; return 0; }
        return (size_t)v->num;
    }
    size_t size() const { return type == Arr ? arr.size() : 0; }
};

struct JParser {
    const char* p;
    const char* end;
    std::string err;

    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }

    bool parse_string(std::string& out) {
        if (p >= end || *p != '"') return fail("json: expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("json: bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - p < 5) return fail("json: bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; ++i) {
                            char c = p[i];
                            cp = cp * 16 + (unsigned)(c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10
                                                      : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0);
                        }
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("json: unterminated string");
        ++p;
        return true;
    }

    bool parse(JVal& v, int depth = 0) {
        if (depth > 64) return fail("json: nesting too deep");
        ws();
        if (p >= end) return fail("json: unexpected end");
        if (*p == '{') {
            v.type = JVal::Obj;
            ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("json: expected ':'");
                ++p;
                JVal child;
                if (!parse(child, depth + 1)) return false;
                v.obj.emplace_back(std::move(key), std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("json: expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.type = JVal::Arr;
            ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JVal child;
                if (!parse(child, depth + 1)) return false;
                v.arr.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("json: expected ',' or ']'");
            }
        }
        if (*p == '"') { v.type = JVal::Str; return parse_string(v.str); }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { v.type = JVal::Bool; v.b = true; p += 4; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { v.type = JVal::Bool; v.b = false; p += 5; return true; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { v.type = JVal::Null; p += 4; return true; }
        {
            std::string num;
            while (p < end && (strchr("+-.eE", *p) || (*p >= '0' && *p <= '9'))) num += *p++;
            if (num.empty()) return fail("json: unexpected character");
            v.type = JVal::Num;
            v.num = strtod(num.c_str(), nullptr);
            return true;
        }
    }
};

// ---------------------------------------------------------------- PNG
struct Image { uint32_t w = 0, h = 0; std::vector<uint8_t> rgba; };

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Returns WR_OK, WR_ERR_UNSUPPORTED (a layout the reference rejects or this decoder does not
// handle) or WR_ERR_INVALID_ARGUMENT (corrupt data).
int decode_png(const uint8_t* data, size_t n, Image& out, std::string& err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(data, sig, 8)) { err = "not a PNG"; return WR_ERR_INVALID_ARGUMENT; }
    size_t off = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_trns = false;
    while (off + 12 <= n) {
        uint32_t len = be32(data + off);
        const uint8_t* type = data + off + 4;
        const uint8_t* body = data + off + 8;
        if ((size_t)len > n - off - 12) { err = "png: truncated chunk"; return WR_ERR_INVALID_ARGUMENT; }
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len); have_trns = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        off += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 16384 || h > 16384) { err = "png: bad size"; return WR_ERR_INVALID_ARGUMENT; }
    if (interlace) { err = "png: interlaced images are not supported"; return WR_ERR_UNSUPPORTED; }
    int channels;
    if (ctype == 2) channels = 3;
    else if (ctype == 6) channels = 4;
    else if (ctype == 3) channels = 1;
    else { err = "png: grey images have 1 or 2 channels; the reference asserts 3 or 4 (resources.rs:51)"; return WR_ERR_UNSUPPORTED; }
    if (ctype == 3 ? !(depth == 1 || depth == 2 || depth == 4 || depth == 8) : depth != 8) {
        err = "png: only 8-bit channels are supported (16-bit formats panic in the reference, resources.rs:138-151)";
        return WR_ERR_UNSUPPORTED;
    }
    const size_t bits = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bits + 7) / 8;
    const size_t bpp = std::max<size_t>(1, bits / 8);
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) { err = "png: inflate failed"; return WR_ERR_INVALID_ARGUMENT; }
    std::vector<uint8_t> img(stride * (size_t)h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* src = raw.data() + (stride + 1) * y;
        uint8_t* dst = img.data() + stride * y;
        const uint8_t* up = y ? dst - stride : nullptr;
        const int filter = src[0];
        ++src;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= bpp ? dst[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[x];
            switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: err = "png: bad filter"; return WR_ERR_INVALID_ARGUMENT;
            }
            dst[x] = (uint8_t)v;
        }
    }
    out.w = w; out.h = h;
    out.rgba.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = img.data() + stride * y;
        uint8_t* o = out.rgba.data() + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; ++x, o += 4) {
            if (ctype == 6) {
                memcpy(o, row + 4 * (size_t)x, 4);
            } else if (ctype == 2) {
                const uint8_t* s = row + 3 * (size_t)x;
                o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255;
                // a colour key expands to an alpha channel, as the `image` crate does
                if (have_trns && trns.size() >= 6 && s[0] == trns[1] && s[1] == trns[3] && s[2] == trns[5]) o[3] = 0;
            } else {
                unsigned idx;
                if (depth == 8) idx = row[x];
                else {
                    const unsigned per = 8u / (unsigned)depth, shift = (per - 1 - (x % per)) * (unsigned)depth;
                    idx = (row[x / per] >> shift) & ((1u << depth) - 1u);
                }
                if ((size_t)idx * 3 + 2 >= plte.size()) { err = "png: palette index out of range"; return WR_ERR_INVALID_ARGUMENT; }
                o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2];
                o[3] = (have_trns && idx < trns.size()) ? trns[idx] : 255;
            }
        }
    }
    return WR_OK;
}

struct MeshData { std::vector<WrModelVertex> verts; std::vector<uint32_t> idx; uint32_t material = 0; };
struct TexData { uint32_t image = 0; int srgb = 0; };
struct MatData { uint32_t base = 0, normal = 0, mr = 0; float metallic = 1.0f, roughness = 1.0f; };

}  // namespace

struct WrGlb {
    std::vector<MeshData> meshes;
    std::vector<Image> images;
    std::vector<TexData> textures;  // (image, srgb) pairs in order of first use by the materials
    std::vector<MatData> materials;
};

namespace {

struct Accessor {
    const uint8_t* base = nullptr;
    size_t stride = 0, count = 0;
    int comp = 0, ncomp = 0;
    bool normalized = false;
};

size_t comp_size(int c) { return c == 5120 || c == 5121 ? 1 : c == 5122 || c == 5123 ? 2 : c == 5125 || c == 5126 ? 4 : 0; }

bool get_accessor(const JVal& js, const std::vector<uint8_t>& bin, long index, Accessor& a, std::string& err) {
    const JVal* accs = js.get("accessors");
    const JVal* views = js.get("bufferViews");
    if (!accs || index < 0 || (size_t)index >= accs->size() || !views) { err = "glb: bad accessor index"; return false; }
    const JVal& acc = accs->arr[(size_t)index];
    if (acc.has("sparse")) { err = "glb: sparse accessors are not supported"; return false; }
    long bvi = acc.integer("bufferView", -1);
    if (bvi < 0 || (size_t)bvi >= views->size()) { err = "glb: accessor without a buffer view"; return false; }
    const JVal& bv = views->arr[(size_t)bvi];
    if (bv.integer("buffer", 0) != 0) { err = "glb: only the embedded buffer 0 is supported"; return false; }
    a.comp = (int)acc.integer("componentType", 0);
    const JVal* ty = acc.get("type");
    const std::string t = ty ? ty->str : "";
    a.ncomp = t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : 0;
    const size_t cs = comp_size(a.comp);
    if (!cs || !a.ncomp) { err = "glb: unsupported accessor type"; return false; }
    bool ok = true;
    a.count = acc.unsigned_integer("count", 0, ok);
    const JVal* nrm = acc.get("normalized");
    a.normalized = nrm && nrm->type == JVal::Bool && nrm->b;
    const size_t view_off = bv.unsigned_integer("byteOffset", 0, ok), acc_off = acc.unsigned_integer("byteOffset", 0, ok);
    const size_t view_len = bv.unsigned_integer("byteLength", 0, ok);
    a.stride = bv.unsigned_integer("byteStride", 0, ok);
    if (!ok) { err = "glb: accessor or buffer view with a negative, fractional or oversized number"; return false; }
    const size_t elem = cs * (size_t)a.ncomp;
    if (a.stride == 0) a.stride = elem;
    if (a.stride < elem) { err = "glb: byteStride smaller than the element"; return false; }
    // all comparisons by subtraction from known-good sizes: nothing here can wrap
    if (view_off > bin.size() || view_len > bin.size() - view_off) { err = "glb: buffer view exceeds the buffer"; return false; }
    if (a.count) {
        if (acc_off > view_len || elem > view_len - acc_off || a.count - 1 > (view_len - acc_off - elem) / a.stride) {
            err = "glb: accessor exceeds its buffer view";
            return false;
        }
    }
    const size_t start = view_off + acc_off;
    a.base = bin.data() + start;
    return true;
}

float comp_float(const Accessor& a, size_t i, int c) {
    const uint8_t* p = a.base + a.stride * i + comp_size(a.comp) * (size_t)c;
    switch (a.comp) {
        case 5126: { float f; memcpy(&f, p, 4); return f; }
        case 5120: { int8_t v; memcpy(&v, p, 1); return a.normalized ? std::max((float)v / 127.0f, -1.0f) : (float)v; }
        case 5121: { uint8_t v = *p; return a.normalized ? (float)v / 255.0f : (float)v; }
        case 5122: { int16_t v; memcpy(&v, p, 2); return a.normalized ? std::max((float)v / 32767.0f, -1.0f) : (float)v; }
        case 5123: { uint16_t v; memcpy(&v, p, 2); return a.normalized ? (float)v / 65535.0f : (float)v; }
        case 5125: { uint32_t v; memcpy(&v, p, 4); return a.normalized ? (float)v / 4294967295.0f : (float)v; }
    }
    return 0.0f;
}
uint32_t comp_uint(const Accessor& a, size_t i) {
    const uint8_t* p = a.base + a.stride * i;
    switch (a.comp) {
        case 5121: return *p;
        case 5123: { uint16_t v; memcpy(&v, p, 2); return v; }
        case 5125: { uint32_t v; memcpy(&v, p, 4); return v; }
    }
    return 0;
}

int parse_glb(const uint8_t* data, size_t bytes, WrGlb& g, std::string& err) {
    if (bytes < 20 || memcmp(data, "glTF", 4)) { err = "not a glTF binary"; return WR_ERR_INVALID_ARGUMENT; }
    uint32_t version;
    memcpy(&version, data + 4, 4);
    if (version != 2) { err = "glTF version is not 2"; return WR_ERR_UNSUPPORTED; }
    JVal js;
    std::vector<uint8_t> bin;
    bool have_json = false;
    size_t off = 12;
    while (off + 8 <= bytes) {
        uint32_t clen, ctype;
        memcpy(&clen, data + off, 4);
        memcpy(&ctype, data + off + 4, 4);
        if ((size_t)clen > bytes - off - 8) { err = "glb: truncated chunk"; return WR_ERR_INVALID_ARGUMENT; }
        const uint8_t* body = data + off + 8;
        if (ctype == 0x4E4F534Au) {  // "JSON"
            JParser jp{(const char*)body, (const char*)body + clen, {}};
            if (!jp.parse(js)) { err = jp.err; return WR_ERR_INVALID_ARGUMENT; }
            have_json = true;
        } else if (ctype == 0x004E4942u) {  // "BIN\0"
            bin.assign(body, body + clen);
        }
        off += 8 + (size_t)clen;
    }
    if (!have_json) { err = "glb: no JSON chunk"; return WR_ERR_INVALID_ARGUMENT; }

    // meshes: one per primitive, document order (mesh.rs:20-21)
    if (const JVal* meshes = js.get("meshes")) {
        for (const JVal& mesh : meshes->arr) {
            const JVal* prims = mesh.get("primitives");
            if (!prims) continue;
            for (const JVal& prim : prims->arr) {
                const JVal* attrs = prim.get("attributes");
                if (!attrs || !attrs->has("POSITION") || !attrs->has("NORMAL") || !attrs->has("TEXCOORD_0") || !prim.has("indices")) {
                    err = "glb: a primitive needs POSITION, NORMAL, TEXCOORD_0 and indices (mesh.rs:25-67 unwraps them)";
                    return WR_ERR_UNSUPPORTED;
                }
                Accessor pos, nrm, uv, tan, ind;
                if (!get_accessor(js, bin, attrs->integer("POSITION", -1), pos, err) ||
                    !get_accessor(js, bin, attrs->integer("NORMAL", -1), nrm, err) ||
                    !get_accessor(js, bin, attrs->integer("TEXCOORD_0", -1), uv, err) ||
                    !get_accessor(js, bin, prim.integer("indices", -1), ind, err))
                    return WR_ERR_INVALID_ARGUMENT;
                const bool has_tan = attrs->has("TANGENT");
                if (has_tan && !get_accessor(js, bin, attrs->integer("TANGENT", -1), tan, err)) return WR_ERR_INVALID_ARGUMENT;
                if (pos.ncomp != 3 || nrm.ncomp != 3 || uv.ncomp != 2 || (has_tan && tan.ncomp != 4) || ind.ncomp != 1 ||
                    nrm.count < pos.count || uv.count < pos.count || (has_tan && tan.count < pos.count)) {
                    err = "glb: inconsistent vertex attributes";
                    return WR_ERR_INVALID_ARGUMENT;
                }
                MeshData m;
                m.material = (uint32_t)prim.integer("material", 0);  // mesh.rs:23
                m.verts.resize(pos.count);
                const float nan = std::numeric_limits<float>::quiet_NaN();
                for (size_t i = 0; i < pos.count; ++i) {
                    WrModelVertex& v = m.verts[i];
                    for (int c = 0; c < 3; ++c) { v.position[c] = comp_float(pos, i, c); v.normal[c] = comp_float(nrm, i, c); }
                    v.tex_coords[0] = comp_float(uv, i, 0); v.tex_coords[1] = comp_float(uv, i, 1);
                    if (has_tan) {
                        const float w = comp_float(tan, i, 3);
                        float t[3];
                        for (int c = 0; c < 3; ++c) { t[c] = comp_float(tan, i, c) * w; v.tangent[c] = t[c]; }
                        const float* nn = v.normal;
                        v.bitangent[0] = nn[1] * t[2] - nn[2] * t[1];
                        v.bitangent[1] = nn[2] * t[0] - nn[0] * t[2];
                        v.bitangent[2] = nn[0] * t[1] - nn[1] * t[0];
                    } else {
                        for (int c = 0; c < 3; ++c) { v.tangent[c] = nan; v.bitangent[c] = nan; }
                    }
                }
                m.idx.resize(ind.count);
                for (size_t i = 0; i < ind.count; ++i) {
                    m.idx[i] = comp_uint(ind, i);
                    if (m.idx[i] >= pos.count) { err = "glb: index out of range"; return WR_ERR_INVALID_ARGUMENT; }
                }
                g.meshes.push_back(std::move(m));
            }
        }
    }

    // images (decoded on demand would hide errors the reference raises at import; decode all)
    if (const JVal* images = js.get("images")) {
        const JVal* views = js.get("bufferViews");
        for (const JVal& img : images->arr) {
            long bvi = img.integer("bufferView", -1);
            if (bvi < 0 || !views || (size_t)bvi >= views->size()) { err = "glb: images must be embedded buffer views"; return WR_ERR_UNSUPPORTED; }
            const JVal& bv = views->arr[(size_t)bvi];
            bool ok = true;
            const size_t o = bv.unsigned_integer("byteOffset", 0, ok), len = bv.unsigned_integer("byteLength", 0, ok);
            if (!ok || o > bin.size() || len > bin.size() - o) { err = "glb: image exceeds the buffer"; return WR_ERR_INVALID_ARGUMENT; }
            Image im;
            int rc;
            if (len >= 3 && bin[o] == 0xFF && bin[o + 1] == 0xD8) rc = wr_decode_jpeg(bin.data() + o, len, &im.w, &im.h, im.rgba, err);
            else rc = decode_png(bin.data() + o, len, im, err);
            if (rc != WR_OK) return rc;
            g.images.push_back(std::move(im));
        }
    }

    // materials (resources.rs:36-130): texture slots resolve to image indices, missing -> 0
    std::map<std::pair<uint32_t, int>, uint32_t> cache;
    auto tex_id = [&](uint32_t image, int srgb) {
        auto key = std::make_pair(image, srgb);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        uint32_t id = (uint32_t)g.textures.size();
        g.textures.push_back(TexData{image, srgb});
        cache[key] = id;
        return id;
    };
    const JVal* textures = js.get("textures");
    auto source = [&](const JVal* texinfo) -> long {
        if (!texinfo || !textures) return 0;
        long ti = texinfo->integer("index", -1);
        if (ti < 0 || (size_t)ti >= textures->size()) return -1;
        return textures->arr[(size_t)ti].integer("source", 0);
    };
    if (const JVal* mats = js.get("materials")) {
        for (const JVal& m : mats->arr) {
            const JVal* pbr = m.get("pbrMetallicRoughness");
            JVal empty;
            empty.type = JVal::Obj;
            if (!pbr) pbr = &empty;
            long sb = source(pbr->get("baseColorTexture")), sn = source(m.get("normalTexture")), sm = source(pbr->get("metallicRoughnessTexture"));
            for (long s : {sb, sn, sm})
                if (s < 0 || (size_t)s >= g.images.size()) { err = "glb: material names a missing image"; return WR_ERR_INVALID_ARGUMENT; }
            MatData md;
            md.base = tex_id((uint32_t)sb, 1);
            md.normal = tex_id((uint32_t)sn, 0);
            md.mr = tex_id((uint32_t)sm, 0);
            md.metallic = (float)pbr->number("metallicFactor", 1.0);
            md.roughness = (float)pbr->number("roughnessFactor", 1.0);
            g.materials.push_back(md);
        }
    }
    return WR_OK;
}

void set_err(char* err, size_t cap, const std::string& m) {
    if (err && cap) { strncpy(err, m.c_str(), cap - 1); err[cap - 1] = 0; }
}

}  // namespace

extern "C" {

int wr_glb_parse(const void* data, size_t bytes, WrGlb** out, char* err, size_t err_cap) {
    if (!data || !out) return WR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    // nothing may leave through the C boundary: allocation failures and anything else the
    // parser or the decoders throw become status codes
    WrGlb* g = nullptr;
    try {
        g = new WrGlb();
        std::string e;
        int rc = parse_glb((const uint8_t*)data, bytes, *g, e);
        if (rc != WR_OK) { set_err(err, err_cap, e); delete g; return rc; }
        *out = g;
        return WR_OK;
    } catch (const std::bad_alloc&) {
        delete g;
        set_err(err, err_cap, "glb: out of memory");
        return WR_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& ex) {
        delete g;
        set_err(err, err_cap, std::string("glb: ") + ex.what());
        return WR_ERR_INVALID_ARGUMENT;
    } catch (...) {
        delete g;
        set_err(err, err_cap, "glb: unexpected failure");
        return WR_ERR_INVALID_ARGUMENT;
    }
}

int wr_glb_load_file(const char* path, WrGlb** out, char* err, size_t err_cap) {
    if (!path || !out) return WR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, err_cap, std::string("cannot open ") + path); return WR_ERR_INVALID_ARGUMENT; }
    std::vector<uint8_t> buf;
    try {
        uint8_t tmp[1 << 16];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    } catch (...) {
        fclose(f);
        set_err(err, err_cap, "glb: out of memory");
        return WR_ERR_OUT_OF_MEMORY;
    }
    fclose(f);
    return wr_glb_parse(buf.data(), buf.size(), out, err, err_cap);
}

void wr_glb_free(WrGlb* g) { delete g; }

void wr_glb_counts(const WrGlb* g, uint32_t* meshes, uint32_t* materials, uint32_t* textures) {
    if (meshes) *meshes = g ? (uint32_t)g->meshes.size() : 0;
    if (materials) *materials = g ? (uint32_t)g->materials.size() : 0;
    if (textures) *textures = g ? (uint32_t)g->textures.size() : 0;
}

int wr_glb_mesh(const WrGlb* g, uint32_t i, const WrModelVertex** vertices, uint32_t* vertex_count, const uint32_t** indices,
                uint32_t* index_count, uint32_t* material) {
    if (!g || i >= g->meshes.size()) return WR_ERR_INVALID_ARGUMENT;
    const MeshData& m = g->meshes[i];
    if (vertices) *vertices = m.verts.data();
    if (vertex_count) *vertex_count = (uint32_t)m.verts.size();
    if (indices) *indices = m.idx.data();
    if (index_count) *index_count = (uint32_t)m.idx.size();
    if (material) *material = m.material;
    return WR_OK;
}

int wr_glb_texture(const WrGlb* g, uint32_t i, const uint8_t** rgba8, uint32_t* width, uint32_t* height, int* srgb) {
    if (!g || i >= g->textures.size()) return WR_ERR_INVALID_ARGUMENT;
    const Image& im = g->images[g->textures[i].image];
    if (rgba8) *rgba8 = im.rgba.data();
    if (width) *width = im.w;
    if (height) *height = im.h;
    if (srgb) *srgb = g->textures[i].srgb;
    return WR_OK;
}

int wr_glb_material(const WrGlb* g, uint32_t i, uint32_t* base, uint32_t* normal, uint32_t* mr, float* metallic, float* roughness) {
    if (!g || i >= g->materials.size()) return WR_ERR_INVALID_ARGUMENT;
    const MatData& m = g->materials[i];
    if (base) *base = m.base;
    if (normal) *normal = m.normal;
    if (mr) *mr = m.mr;
    if (metallic) *metallic = m.metallic;
    if (roughness) *roughness = m.roughness;
    return WR_OK;
}

int wr_upload_glb(WrContext* ctx, uint32_t model, const WrGlb* g) {
    if (!ctx || !g) return WR_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> tex_ids(g->textures.size(), 0);
    for (size_t i = 0; i < g->textures.size(); ++i) {
        const Image& im = g->images[g->textures[i].image];
        int rc = wr_upload_texture(ctx, im.rgba.data(), im.w, im.h, g->textures[i].srgb, &tex_ids[i]);
        if (rc != WR_OK) return rc;
    }
    for (const MatData& m : g->materials) {
        int rc = wr_upload_material(ctx, model, tex_ids[m.base], tex_ids[m.normal], tex_ids[m.mr], m.metallic, m.roughness, nullptr);
        if (rc != WR_OK) return rc;
    }
    for (const MeshData& m : g->meshes) {
        int rc = wr_upload_mesh(ctx, model, m.verts.data(), (uint32_t)m.verts.size(), m.idx.data(), (uint32_t)m.idx.size(), m.material);
        if (rc != WR_OK) return rc;
    }
    return WR_OK;
}

}  // extern "C"
