// art_glb.hip -- host-side GLB ingest: the step right before the ray-tracing path.
//
// Reproduces GltfModelReader of /root/reference/src/vk_renderer/model_reader/gltf_model_reader.rs on the data the
// `gltf` crate hands it: open (:55-150: one mesh, one buffer, per-primitive attribute windows + material textures),
// normalize_vectors (:415-460), coerce_images_to_format (:463-527) with permute_pixels (:542-573), validate_model
// (:643-681), copy_model_data_to_ptr (:156-281: the 48-byte interleave, indices, texture array) and
// get_primitives_bounding_sphere (:283-399, Ritter).  The crate itself is an un-vendored dependency (gltf = "1.0.0",
// Cargo.toml:40), so its job is restated here: GLB container, glTF JSON, PNG decode (zlib inflate + unfilter).
// No device code in this file; JPEG images go through art_jpeg.hip (baseline and progressive Huffman streams).
#include "../../include/art.h"
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <exception>
#include <string>
#include <vector>

namespace art { bool decode_jpeg(const uint8_t *data, size_t n, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, int &channels, std::string &err); } // art_jpeg.hip
using art::decode_jpeg;

namespace {

thread_local std::string g_glb_err;

// ------------------------------------------------------------------------------------------------ tiny JSON
struct JV {
    enum T { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false; double n = 0; std::string s;
    std::vector<JV> a; std::vector<std::pair<std::string, JV>> o;
    const JV *get(const char *k) const { if (t != Obj) return nullptr; for (auto &kv : o) if (kv.first == k) return &kv.second; return nullptr; }
    bool has(const char *k) const { return get(k) != nullptr; }
    double num(const char *k, double d) const { const JV *v = get(k); return v && v->t == Num ? v->n : d; }
    size_t size() const { return t == Arr ? a.size() : 0; }
};
struct JP {
    const char *p, *e; bool ok = true; int depth = 0;
    static constexpr int kMaxDepth = 64; // glTF documents nest a handful of levels; a crafted file must not recurse the stack away
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
    JV val() {
        JV v;
        if (++depth > kMaxDepth) ok = false;
        if (ok) v = val_body();
        depth--;
        return v;
    }
    JV val_body() {
        JV v; ws();
        if (p >= e) { ok = false; return v; }
        if (*p == '{') {
            v.t = JV::Obj; p++; ws();
            if (p < e && *p == '}') { p++; return v; }
            while (ok) {
                ws(); JV k = val(); if (k.t != JV::Str) { ok = false; break; }
                ws(); if (p >= e || *p != ':') { ok = false; break; } p++;
                v.o.emplace_back(k.s, val());
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; break; }
                ok = false;
            }
        } else if (*p == '[') {
            v.t = JV::Arr; p++; ws();
            if (p < e && *p == ']') { p++; return v; }
            while (ok) {
                v.a.push_back(val());
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; break; }
                ok = false;
            }
        } else if (*p == '"') {
            v.t = JV::Str; p++;
            while (p < e && *p != '"') {
                if (*p == '\\' && p + 1 < e) {
                    p++;
                    switch (*p) { case 'n': v.s += '\n'; break; case 't': v.s += '\t'; break; case 'r': v.s += '\r'; break; case 'b': v.s += '\b'; break; case 'f': v.s += '\f'; break;
                                  case 'u': if (e - p < 5) { ok = false; return v; } v.s += '?'; p += 4; break; default: v.s += *p; }
                    p++;
                } else v.s += *p++;
            }
            if (p < e) p++; else ok = false;
        } else if (!std::strncmp(p, "true", 4)) { v.t = JV::Bool; v.b = true; p += 4; }
        else if (!std::strncmp(p, "false", 5)) { v.t = JV::Bool; p += 5; }
        else if (!std::strncmp(p, "null", 4)) { p += 4; }
        else { char *q = nullptr; v.t = JV::Num; v.n = std::strtod(p, &q); if (q == p) ok = false; p = q; }
        return v;
    }
};

// ------------------------------------------------------------------------------------------------ PNG -> gltf::image::Data
enum ImgFormat { F_R8 = 0, F_R8G8, F_R8G8B8, F_R8G8B8A8, F_B8G8R8, F_B8G8R8A8, F_R16, F_R16G16, F_R16G16B16, F_R16G16B16A16 }; // gltf::image::Format
struct Image { std::vector<uint8_t> pixels; int format = F_R8; uint32_t width = 0, height = 0; };

constexpr uint32_t kMaxImageEdge = 16384;
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool decode_png(const uint8_t *d, size_t n, Image &out, std::string &err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 || std::memcmp(d, sig, 8)) { err = "not a PNG stream"; return false; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t p = 8;
    while (p + 12 <= n) {
        uint32_t len = be32(d + p); const uint8_t *typ = d + p + 4, *data = d + p + 8;
        if (p + 12 + len > n) { err = "truncated PNG chunk"; return false; }
        if (!std::memcmp(typ, "IHDR", 4) && len >= 13) { w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!std::memcmp(typ, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(typ, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(typ, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(typ, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (!w || !h) { err = "PNG without IHDR"; return false; }
    if (w > kMaxImageEdge || h > kMaxImageEdge || (size_t)w * h > ((size_t)1 << 26)) { err = "PNG extent beyond 16384 / 64 Mpixel"; return false; }
    if (interlace) { err = "interlaced PNG not supported"; return false; }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || (depth != 8 && depth != 16 && !(ctype == 3 && (depth == 1 || depth == 2 || depth == 4)) && !(ctype == 0 && (depth == 1 || depth == 2 || depth == 4)))) { err = "unsupported PNG colour type / bit depth"; return false; }
    size_t bpp_bits = (size_t)ch * depth, stride = (w * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) { err = "PNG inflate failed"; return false; }
    std::vector<uint8_t> img(stride * (size_t)h);
    for (uint32_t y = 0; y < h; y++) { // unfilter (PNG spec 9.2)
        const uint8_t *src = raw.data() + (stride + 1) * (size_t)y; uint8_t ft = src[0]; src++;
        uint8_t *cur = img.data() + stride * (size_t)y; const uint8_t *up = y ? cur - stride : nullptr;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, v = src[i];
            switch (ft) {
                case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break;
                case 4: { int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: err = "bad PNG filter type"; return false;
            }
            cur[i] = (uint8_t)v;
        }
    }
    out.width = w; out.height = h;
    size_t npx = (size_t)w * h;
    auto sample = [&](uint32_t x, uint32_t y) -> uint32_t { // sub-byte samples, MSB first
        const uint8_t *row = img.data() + stride * (size_t)y; size_t bit = (size_t)x * depth;
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
    };
    if (ctype == 3) { // the image crate expands palettes to RGB8 (RGBA8 with tRNS)
        bool alpha = !trns.empty();
        out.format = alpha ? F_R8G8B8A8 : F_R8G8B8;
        out.pixels.resize(npx * (alpha ? 4 : 3));
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++) {
                uint32_t idx = depth == 8 ? img[stride * (size_t)y + x] : sample(x, y);
                if ((size_t)idx * 3 + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; }
                uint8_t *o = out.pixels.data() + ((size_t)y * w + x) * (alpha ? 4 : 3);
                o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2];
                if (alpha) o[3] = idx < trns.size() ? trns[idx] : 255;
            }
        return true;
    }
    if (depth < 8) { // grey 1/2/4 bit -> L8 scaled
        out.format = F_R8; out.pixels.resize(npx);
        uint32_t mx = (1u << depth) - 1u;
        for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) out.pixels[(size_t)y * w + x] = (uint8_t)(sample(x, y) * 255u / mx);
        return true;
    }
    if (depth == 8) { out.format = ch == 1 ? F_R8 : ch == 2 ? F_R8G8 : ch == 3 ? F_R8G8B8 : F_R8G8B8A8; out.pixels = std::move(img); return true; }
    out.format = ch == 1 ? F_R16 : ch == 2 ? F_R16G16 : ch == 3 ? F_R16G16B16 : F_R16G16B16A16; // 16-bit: PNG is big-endian, the crate stores native (little) endian
    out.pixels.resize(img.size());
    for (size_t i = 0; i + 1 < img.size(); i += 2) { out.pixels[i] = img[i + 1]; out.pixels[i + 1] = img[i]; }
    return true;
}

// ------------------------------------------------------------------------------------------------ reader state
enum { A_VERTICES = 1, A_TEX_COORDS = 2, A_NORMALS = 4, A_TANGENTS = 8, A_INDICES = 16 }; // MeshAttributeType, model_reader.rs:6-12
enum { T_ALBEDO = 1, T_ORM = 2, T_NORMAL = 4, T_EMISSIVE = 8 };                            // TextureType, model_reader.rs:14-19
struct Attr { uint64_t start = 0, len = 0; uint32_t elem_size = 0, stride = 0; uint64_t count() const { return stride ? len / stride : 0; } }; // :10-34
struct Prim { std::map<int, Attr> attrs; std::map<int, int> textures; /* type -> image index */ };

} // namespace

struct ArtGlb {
    std::vector<uint8_t> buffer;
    std::vector<Image> images;
    std::vector<Prim> prims;
};

namespace {

int32_t gfail(int32_t code, const std::string &m) { g_glb_err = m; return code; }

int type_components(const std::string &t) { return t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : t == "MAT2" ? 4 : t == "MAT3" ? 9 : t == "MAT4" ? 16 : 0; }
int component_bytes(int ct) { return ct == 5120 || ct == 5121 ? 1 : ct == 5122 || ct == 5123 ? 2 : ct == 5125 || ct == 5126 ? 4 : 0; }

// a JSON number that is a non-negative integer a double holds exactly (anything else -- negative, fractional, NaN, 1e300 -- is refused
// before it is cast: the cast of such a double is undefined behaviour, and a wrapped offset passes every later check)
bool json_u64(const JV *v, uint64_t &out) {
    if (!v || v->t != JV::Num || !(v->n >= 0.0) || !(v->n <= 9007199254740992.0) || v->n != std::floor(v->n)) return false;
    out = (uint64_t)v->n;
    return true;
}
// obj[key] as an index into a table of `limit` entries; false when absent, not an integer or out of range
bool json_index(const JV &obj, const char *key, size_t limit, size_t &out) {
    uint64_t v = 0;
    if (!json_u64(obj.get(key), v) || v >= limit) return false;
    out = (size_t)v;
    return true;
}
// optional non-negative integer member with a default
bool json_opt_u64(const JV &obj, const char *key, uint64_t dflt, uint64_t &out) {
    if (!obj.has(key)) { out = dflt; return true; }
    return json_u64(obj.get(key), out);
}

// get_mesh_attribute_from_accessor, :403-412.  The window [start, start + (count - 1) * stride + elem_size) is checked against the
// binary chunk here, in 64-bit arithmetic that cannot wrap, so that every later reader (normalize_vectors, copy_model_data,
// bounding_sphere) may index it freely.  The reference indexes Rust slices and panics on such files; this returns an error.
bool accessor_attr(const JV &doc, size_t acc_idx, uint64_t buffer_size, Attr &out, std::string &err) {
    const JV *accs = doc.get("accessors"), *views = doc.get("bufferViews");
    if (!accs || acc_idx >= accs->size()) { err = "accessor index out of range"; return false; }
    const JV &a = accs->a[acc_idx];
    size_t vi = 0;
    if (!views || !json_index(a, "bufferView", views->size(), vi)) { err = "accessor without a buffer view"; return false; }
    const JV &v = views->a[vi];
    const JV *ty = a.get("type");
    uint64_t ct = 0;
    if (!json_u64(a.get("componentType"), ct) || ct > 65535) { err = "unsupported accessor type"; return false; }
    const uint64_t size = (uint64_t)type_components(ty && ty->t == JV::Str ? ty->s : "") * (uint64_t)component_bytes((int)ct);
    if (!size) { err = "unsupported accessor type"; return false; }
    uint64_t stride = 0, a_off = 0, v_off = 0, count = 0;
    if (!json_opt_u64(v, "byteStride", size, stride) || !json_opt_u64(a, "byteOffset", 0, a_off) || !json_opt_u64(v, "byteOffset", 0, v_off) || !json_u64(a.get("count"), count)) {
        err = "accessor: byteStride / byteOffset / count must be non-negative integers"; return false;
    }
    if (stride < size || stride > 65536) { err = "accessor: byteStride smaller than the element (or absurdly large)"; return false; }
    if (count == 0) { err = "accessor: count is zero"; return false; }
    const uint64_t start = a_off + v_off; // both <= 2^53
    if (start > buffer_size || size > buffer_size - start || count - 1 > (buffer_size - start - size) / stride) { err = "accessor window out of range"; return false; }
    out.start = start; out.len = count * stride; // count <= buffer_size: no overflow
    out.elem_size = (uint32_t)size; out.stride = (uint32_t)stride;
    return true;
}

// the element sizes validate_model (:643-681) insists on
uint32_t attr_want_size(int type) { return type == A_VERTICES ? 12 : type == A_TEX_COORDS ? 8 : type == A_NORMALS ? 12 : type == A_TANGENTS ? 16 : 0; }

// generate_src_to_dst_map (:529-540) on channel-position arrays: map[src byte] = dst byte or -1
void src_to_dst_map(const int src_pos[4], const int dst_pos[4], int map[4]) { // index: r g b a; value: byte position or -1
    for (int i = 0; i < 4; i++) map[i] = -1;
    for (int c = 0; c < 4; c++) if (src_pos[c] >= 0 && dst_pos[c] >= 0) map[src_pos[c]] = dst_pos[c];
}

// nothing unwinds across the C ABI: a file that asks for more memory than there is (huge image extents, vertex counts) is an error code
template <class F> int32_t glb_guard(const char *who, F f) {
    try { return f(); }
    catch (const std::bad_alloc &) { return gfail(ART_E_NOMEM, std::string(who) + ": out of memory"); }
    catch (const std::exception &e) { return gfail(ART_E_INVALID, std::string(who) + ": " + e.what()); }
    catch (...) { return gfail(ART_E_INVALID, std::string(who) + ": unexpected failure"); }
}

} // namespace

extern "C" {

const char *art_glb_last_error(void) { return g_glb_err.c_str(); }

// permute_pixels (:542-573): out texel byte map[s] <- src texel byte s; unmapped destination bytes stay 0
int32_t art_glb_permute_pixels(const uint8_t *src, size_t src_len, uint32_t src_texel, const int32_t *map, uint32_t map_len, uint32_t dst_texel, uint8_t *dst, size_t dst_cap) {
    if (!src || !map || !dst || !src_texel || !dst_texel) return gfail(ART_E_INVALID, "art_glb_permute_pixels: bad argument");
    size_t n = src_len / src_texel;
    if (dst_cap < n * dst_texel) return gfail(ART_E_INVALID, "art_glb_permute_pixels: destination too small");
    std::memset(dst, 0, n * dst_texel);
    for (size_t t = 0; t < n; t++)
        for (uint32_t s = 0; s < src_texel && s < map_len; s++)
            if (map[s] >= 0 && (uint32_t)map[s] < dst_texel) dst[t * dst_texel + map[s]] = src[t * src_texel + s];
    return ART_OK;
}

static int32_t glb_open_impl(const char *path, int32_t normalize_vectors, int32_t coerce_format, ArtGlb **out) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return gfail(ART_E_INVALID, std::string("Could not read file ") + path);
    std::vector<uint8_t> file;
    { uint8_t buf[65536]; size_t r; while ((r = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + r); }
    std::fclose(f);
    auto le32 = [&](size_t o) { return (uint32_t)file[o] | ((uint32_t)file[o + 1] << 8) | ((uint32_t)file[o + 2] << 16) | ((uint32_t)file[o + 3] << 24); };
    if (file.size() < 20 || le32(0) != 0x46546C67u || le32(4) != 2u) return gfail(ART_E_INVALID, "not a GLB 2.0 container");
    size_t off = 12; std::string json; std::unique_ptr<ArtGlb> g(new ArtGlb());
    bool have_bin = false;
    while (off + 8 <= file.size()) {
        uint32_t len = le32(off), typ = le32(off + 4);
        if (off + 8 + (size_t)len > file.size()) return gfail(ART_E_INVALID, "truncated GLB chunk");
        if (typ == 0x4E4F534Au) json.assign((const char *)file.data() + off + 8, len);
        else if (typ == 0x004E4942u && !have_bin) { g->buffer.assign(file.begin() + off + 8, file.begin() + off + 8 + len); have_bin = true; }
        off += 8 + (size_t)len;
    }
    JP jp{json.data(), json.data() + json.size()};
    JV doc = jp.val();
    if (!jp.ok || doc.t != JV::Obj) return gfail(ART_E_INVALID, "glTF JSON does not parse");
    const JV *meshes = doc.get("meshes"), *buffers = doc.get("buffers");
    if (!meshes || meshes->size() != 1) return gfail(ART_E_INVALID, "expected exactly one mesh (gltf_model_reader.rs:62)");
    if (!buffers || buffers->size() != 1 || !have_bin) return gfail(ART_E_INVALID, "expected exactly one buffer (gltf_model_reader.rs:63)");
    // images: decoded up front like gltf::import
    if (const JV *imgs = doc.get("images")) {
        const JV *views = doc.get("bufferViews");
        for (const JV &im : imgs->a) {
            Image I; std::string err;
            size_t vi = 0;
            if (!views || !json_index(im, "bufferView", views->size(), vi)) return gfail(ART_E_INVALID, "image without a buffer view (external URIs are not read)");
            uint64_t o = 0, l = 0;
            if (!json_opt_u64(views->a[vi], "byteOffset", 0, o) || !json_u64(views->a[vi].get("byteLength"), l)) return gfail(ART_E_INVALID, "image buffer view: byteOffset / byteLength must be non-negative integers");
            if (o > g->buffer.size() || l > g->buffer.size() - o) return gfail(ART_E_INVALID, "image buffer view out of range");
            const uint8_t *img = g->buffer.data() + o;
            if (l >= 2 && img[0] == 0xFF && img[1] == 0xD8) { // JPEG: RGB8 or R8, like the image crate's decode of it
                int ch = 0;
                if (!decode_jpeg(img, (size_t)l, I.pixels, I.width, I.height, ch, err)) return gfail(ART_E_INVALID, "image decode: " + err);
                if (!I.width || !I.height || I.pixels.size() != (size_t)I.width * I.height * (size_t)ch) return gfail(ART_E_INVALID, "image decode: JPEG extent and pixel data disagree");
                I.format = ch == 1 ? F_R8 : F_R8G8B8;
            } else if (!decode_png(img, (size_t)l, I, err)) return gfail(ART_E_INVALID, "image decode: " + err);
            g->images.push_back(std::move(I));
        }
    }
    const JV *prims = meshes->a[0].get("primitives");
    const JV *materials = doc.get("materials"), *textures = doc.get("textures");
    for (size_t pi = 0; prims && pi < prims->size(); pi++) {
        const JV &pd = prims->a[pi];
        Prim P; std::string err;
        const uint64_t bsz = g->buffer.size();
        uint64_t ai = 0;
        if (pd.has("indices")) { Attr a; if (!json_u64(pd.get("indices"), ai) || !accessor_attr(doc, (size_t)ai, bsz, a, err)) return gfail(ART_E_INVALID, err.empty() ? "accessor index out of range" : err); P.attrs[A_INDICES] = a; }
        if (const JV *at = pd.get("attributes"))
            for (auto &kv : at->o) {
                int ty = kv.first == "POSITION" ? A_VERTICES : kv.first == "NORMAL" ? A_NORMALS : kv.first == "TANGENT" ? A_TANGENTS : kv.first == "TEXCOORD_0" ? A_TEX_COORDS : 0;
                if (!ty) continue;
                Attr a; if (!json_u64(&kv.second, ai) || !accessor_attr(doc, (size_t)ai, bsz, a, err)) return gfail(ART_E_INVALID, err.empty() ? "accessor index out of range" : err);
                // validate_model's element sizes (:643-681), checked HERE: normalize_vectors below reads and writes 12 bytes per POSITION element
                if (a.elem_size != attr_want_size(ty)) return gfail(ART_E_INVALID, "validate_model: attribute element size");
                P.attrs[ty] = a;
            }
        size_t mi = 0;
        if (materials && json_index(pd, "material", materials->size(), mi)) {
            const JV &m = materials->a[mi];
            const JV *pbr = m.get("pbrMetallicRoughness");
            auto tex_image = [&](const JV *info) -> int {
                if (!info) return -1;
                size_t ti = 0, src = 0;
                if (!textures || !json_index(*info, "index", textures->size(), ti)) return -1;
                if (!json_index(textures->a[ti], "source", g->images.size(), src)) return -1;
                return (int)src;
            };
            const std::pair<int, const JV *> slots[4] = {{T_ALBEDO, pbr ? pbr->get("baseColorTexture") : nullptr}, {T_ORM, pbr ? pbr->get("metallicRoughnessTexture") : nullptr},
                                                          {T_NORMAL, m.get("normalTexture")}, {T_EMISSIVE, m.get("emissiveTexture")}};
            for (auto &sl : slots)
                if (sl.second) {
                    int ii = tex_image(sl.second);
                    if (ii < 0 || (size_t)ii >= g->images.size()) return gfail(ART_E_INVALID, "Cannot open texture idx " + std::to_string(ii));
                    P.textures[sl.first] = ii;
                }
        }
        g->prims.push_back(std::move(P));
    }
    // normalize_vectors (:415-460): positions of every primitive divided by the largest magnitude, if that exceeds 1
    if (normalize_vectors) {
        float max_mag = 1.0f;
        for (auto &P : g->prims) { auto it = P.attrs.find(A_VERTICES); if (it == P.attrs.end()) continue; const Attr &a = it->second;
            for (uint64_t i = 0; i < a.count(); i++) { float v[3]; std::memcpy(v, g->buffer.data() + a.start + i * a.stride, 12); float m = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); if (m > max_mag) max_mag = m; } }
        for (auto &P : g->prims) { auto it = P.attrs.find(A_VERTICES); if (it == P.attrs.end()) continue; const Attr &a = it->second;
            for (uint64_t i = 0; i < a.count(); i++) { float v[3]; uint8_t *q = g->buffer.data() + a.start + i * a.stride; std::memcpy(v, q, 12); v[0] /= max_mag; v[1] /= max_mag; v[2] /= max_mag; std::memcpy(q, v, 12); } }
    }
    // coerce_images_to_format (:463-527): 1 = R8G8B8A8, 2 = B8G8R8A8, 3 = B8G8R8
    if (coerce_format) {
        int dst_pos[4], dsz;
        if (coerce_format == 1) { int d[4] = {0, 1, 2, 3}; std::memcpy(dst_pos, d, 16); dsz = 4; }
        else if (coerce_format == 2) { int d[4] = {2, 1, 0, 3}; std::memcpy(dst_pos, d, 16); dsz = 4; }
        else if (coerce_format == 3) { int d[4] = {2, 1, 0, -1}; std::memcpy(dst_pos, d, 16); dsz = 3; }
        else return gfail(ART_E_INVALID, "Unsupported destination format during format coercion");
        for (auto &P : g->prims)
            for (auto &tx : P.textures) {
                Image &I = g->images[tx.second];
                int src_pos[4], ssz;
                if (I.format == F_R8G8B8) { int s[4] = {0, 1, 2, -1}; std::memcpy(src_pos, s, 16); ssz = 3; }
                else if (I.format == F_R8G8B8A8) { int s[4] = {0, 1, 2, 3}; std::memcpy(src_pos, s, 16); ssz = 4; }
                else if (I.format == F_B8G8R8A8) { int s[4] = {2, 1, 0, 3}; std::memcpy(src_pos, s, 16); ssz = 4; }
                else return gfail(ART_E_INVALID, "Unsupported source format during format coercion");
                int map[4]; src_to_dst_map(src_pos, dst_pos, map);
                bool differs = ssz != dsz;
                for (int s = 0; s < 4; s++) if (map[s] >= 0 && map[s] != s) differs = true;
                if (!differs) continue;
                std::vector<uint8_t> nd((I.pixels.size() / ssz) * dsz);
                int32_t m32[4] = {map[0], map[1], map[2], map[3]};
                art_glb_permute_pixels(I.pixels.data(), I.pixels.size(), (uint32_t)ssz, m32, 4, (uint32_t)dsz, nd.data(), nd.size());
                I.pixels = std::move(nd);
                if (coerce_format == 1) I.format = F_R8G8B8A8; else if (coerce_format == 2) I.format = F_B8G8R8A8;
                else return gfail(ART_E_INVALID, "Unsupported destination format conversion"); // the reference panics here for B8G8R8 (:520)
            }
    }
    // validate_model (:643-681)
    for (auto &P : g->prims) {
        long common = -1;
        for (auto &kv : P.attrs) {
            uint32_t want = attr_want_size(kv.first);
            if (!want) continue;
            if (kv.second.elem_size != want) return gfail(ART_E_INVALID, "validate_model: attribute element size");
            if (common < 0) common = (long)kv.second.count(); else if (common != (long)kv.second.count()) return gfail(ART_E_INVALID, "validate_model: attribute element counts differ");
        }
        int fmt = -1; uint32_t w = 0, h = 0;
        for (auto &tx : P.textures) { const Image &I = g->images[tx.second];
            if (fmt < 0) { fmt = I.format; w = I.width; h = I.height; } else if (w != I.width || h != I.height || fmt != I.format) return gfail(ART_E_INVALID, "validate_model: textures of a primitive differ in extent or format"); }
    }
    *out = g.release();
    return ART_OK;
}

int32_t art_glb_open(const char *path, int32_t normalize_vectors, int32_t coerce_format, ArtGlb **out) {
    if (!path || !out) return gfail(ART_E_INVALID, "art_glb_open: null argument");
    *out = nullptr;
    return glb_guard("art_glb_open", [&] { return glb_open_impl(path, normalize_vectors, coerce_format, out); });
}

int32_t art_glb_close(ArtGlb *g) { delete g; return ART_OK; }

int32_t art_glb_primitive_count(ArtGlb *g, uint32_t *n) { if (!g || !n) return gfail(ART_E_INVALID, "art_glb_primitive_count: null argument"); *n = (uint32_t)g->prims.size(); return ART_OK; }

// copy_model_data_to_ptr (:156-281).  dst == NULL: sizing pass.  infos: one ArtGlbCopyInfo per primitive (may be NULL).
static int32_t glb_copy_impl(ArtGlb *g, uint32_t attr_mask, uint32_t tex_mask, void *dst, size_t cap, ArtGlbCopyInfo *infos, uint32_t n_infos, size_t *total) {
    if (!g) return gfail(ART_E_INVALID, "art_glb_copy_model_data: null reader");
    if (infos && n_infos < g->prims.size()) return gfail(ART_E_INVALID, "art_glb_copy_model_data: infos too short");
    std::vector<int> mesh_flags, tex_flags;
    for (int b = 1; b <= 8; b <<= 1) if (attr_mask & b) mesh_flags.push_back(b);      // bitflag_vec! order, INDICES popped (:162-166)
    for (int b = 1; b <= 8; b <<= 1) if (tex_mask & b) tex_flags.push_back(b);
    size_t written = 0; uint8_t *d = (uint8_t *)dst;
    auto put = [&](const uint8_t *src, size_t n) -> bool { if (d) { if (written + n > cap) return false; std::memcpy(d + written, src, n); } written += n; return true; };
    for (size_t pi = 0; pi < g->prims.size(); pi++) {
        const Prim &P = g->prims[pi];
        ArtGlbCopyInfo ci; std::memset(&ci, 0, sizeof(ci));
        if (!mesh_flags.empty()) {
            ci.mesh_buffer_offset = written;
            auto first = P.attrs.find(mesh_flags[0]);
            if (first == P.attrs.end()) return gfail(ART_E_INVALID, "Mesh attribute " + std::to_string(mesh_flags[0]) + " not found");
            uint64_t count = first->second.count();
            for (uint64_t i = 0; i < count; i++)
                for (int fl : mesh_flags) {
                    auto it = P.attrs.find(fl);
                    if (it == P.attrs.end()) return gfail(ART_E_INVALID, "Mesh attribute " + std::to_string(fl) + " not found");
                    if (!put(g->buffer.data() + it->second.start + i * it->second.stride, it->second.elem_size)) return gfail(ART_E_INVALID, "art_glb_copy_model_data: destination too small");
                }
            ci.mesh_size = written - ci.mesh_buffer_offset;
            ci.single_mesh_element_size = count ? (uint32_t)(ci.mesh_size / count) : 0;
        }
        if (attr_mask & A_INDICES) {
            ci.indices_buffer_offset = written;
            auto it = P.attrs.find(A_INDICES);
            if (it == P.attrs.end()) return gfail(ART_E_INVALID, "Attribute INDICES not found in model");
            ci.indices_size = it->second.count() * it->second.elem_size; ci.single_index_size = it->second.elem_size;
            for (uint64_t i = 0; i < it->second.count(); i++)
                if (!put(g->buffer.data() + it->second.start + i * it->second.stride, it->second.elem_size)) return gfail(ART_E_INVALID, "art_glb_copy_model_data: destination too small");
        }
        if (!tex_flags.empty()) {
            auto ft = P.textures.find(tex_flags[0]);
            if (ft == P.textures.end()) return gfail(ART_E_INVALID, "Texture type " + std::to_string(tex_flags[0]) + " not found in model");
            const Image &I0 = g->images[ft->second];
            ci.image_width = I0.width; ci.image_height = I0.height;
            if (!I0.width || !I0.height) return gfail(ART_E_INVALID, "art_glb_copy_model_data: empty image");
            size_t comp = I0.pixels.size() / ((size_t)I0.width * I0.height);
            if (!comp) return gfail(ART_E_INVALID, "art_glb_copy_model_data: image without pixel data");
            // align_offset (model_reader.rs:144-146) computes comp * ceil(written as f32 / comp as f32): exact below 2^24 bytes, but past 64 MB the
            // f32 cast rounds and the offset can move BACKWARDS into the index data just written (the reference has that latent bug; here the
            // library is its own consumer).  Integer arithmetic: the same value wherever the f32 form is exact.
            written = (written + comp - 1) / comp * comp;
            ci.image_buffer_offset = written; ci.image_mip_levels = 1; ci.image_layers = (uint32_t)tex_flags.size(); ci.image_format = (uint32_t)I0.format;
            for (int tf : tex_flags) {
                auto it = P.textures.find(tf);
                if (it == P.textures.end()) return gfail(ART_E_INVALID, "Texture type " + std::to_string(tf) + " not found in model");
                const Image &I = g->images[it->second];
                if (d && written > cap) return gfail(ART_E_INVALID, "art_glb_copy_model_data: destination too small");
                if (!put(I.pixels.data(), I.pixels.size())) return gfail(ART_E_INVALID, "art_glb_copy_model_data: destination too small");
            }
            ci.image_size = written - ci.image_buffer_offset;
        }
        if (infos) infos[pi] = ci;
    }
    if (total) *total = written;
    return ART_OK;
}

int32_t art_glb_copy_model_data(ArtGlb *g, uint32_t attr_mask, uint32_t tex_mask, void *dst, size_t cap, ArtGlbCopyInfo *infos, uint32_t n_infos, size_t *total) {
    return glb_guard("art_glb_copy_model_data", [&] { return glb_copy_impl(g, attr_mask, tex_mask, dst, cap, infos, n_infos, total); });
}

// get_primitives_bounding_sphere (:283-399): Ritter's two-pass sphere, the reference's arithmetic order
int32_t art_glb_bounding_sphere(ArtGlb *g, float center[3], float *radius) {
    if (!g || !center || !radius) return gfail(ART_E_INVALID, "art_glb_bounding_sphere: null argument");
    const float FMIN = -3.40282347e+38f, FMAX = 3.40282347e+38f;
    float xmax[3] = {FMIN, FMIN, FMIN}, xmin[3] = {FMAX, FMAX, FMAX}, ymin[3] = {FMAX, FMAX, FMAX}, ymax[3] = {FMIN, FMIN, FMIN}, zmin[3] = {FMAX, FMAX, FMAX}, zmax[3] = {FMIN, FMIN, FMIN};
    auto each = [&](auto fn) {
        for (auto &P : g->prims) { auto it = P.attrs.find(A_VERTICES); if (it == P.attrs.end()) continue; const Attr &a = it->second;
            for (uint64_t i = 0; i < a.count(); i++) { float v[3]; std::memcpy(v, g->buffer.data() + a.start + i * a.stride, 12); fn(v); } }
    };
    auto cp = [](float *d, const float *s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; };
    each([&](const float *v) {
        if (v[0] < xmin[0]) cp(xmin, v); if (v[0] > xmax[0]) cp(xmax, v);
        if (v[1] < ymin[1]) cp(ymin, v); if (v[1] > ymax[1]) cp(ymax, v);
        if (v[2] < zmin[2]) cp(zmin, v); if (v[2] > zmax[2]) cp(zmax, v);
    });
    auto d2 = [](const float *a, const float *b) { float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2]; return x * x + y * y + z * z; };
    float xspan = d2(xmax, xmin), yspan = d2(ymax, ymin), zspan = d2(zmax, zmin);
    float dia1[3], dia2[3]; cp(dia1, xmin); cp(dia2, xmax);
    float maxspan = xspan;
    if (yspan > maxspan) { maxspan = yspan; cp(dia1, ymin); cp(dia2, ymax); }
    if (zspan > maxspan) { cp(dia1, zmin); cp(dia2, zmax); }
    float c[3] = {(dia1[0] + dia2[0]) * 0.5f, (dia1[1] + dia2[1]) * 0.5f, (dia1[2] + dia2[2]) * 0.5f};
    float r2 = d2(dia2, c), r = std::sqrt(r2);
    each([&](const float *v) {
        float old2 = d2(v, c);
        if (old2 > r2) {
            float old = std::sqrt(old2);
            r = (r + old) * 0.5f; r2 = r * r;
            float o2n = old - r, recip = 1.0f / old;
            for (int k = 0; k < 3; k++) c[k] = (r * c[k] + o2n * v[k]) * recip;
        }
    });
    cp(center, c); *radius = r;
    return ART_OK;
}

// add_model (renderer.rs:346 -> vk_model.rs:494-528, :508: all four vertex attributes + INDICES, textures ALBEDO|ORM|NORMAL):
// hands every primitive to art_scene_add_primitive in the layout copy_model_data_to_ptr produces
static int32_t glb_add_impl(ArtContext *ctx, ArtGlb *g, const float model3x4[12], uint32_t *first_primitive_id, uint32_t *n_primitives) {
    if (!ctx || !g || !model3x4) return gfail(ART_E_INVALID, "art_scene_add_glb: null argument");
    const uint32_t am = A_VERTICES | A_TEX_COORDS | A_NORMALS | A_TANGENTS | A_INDICES, tm = T_ALBEDO | T_ORM | T_NORMAL;
    size_t total = 0;
    std::vector<ArtGlbCopyInfo> infos(g->prims.size());
    int32_t r = art_glb_copy_model_data(g, am, tm, nullptr, 0, infos.data(), (uint32_t)infos.size(), &total);
    if (r) return r;
    std::vector<uint8_t> blob(total);
    r = art_glb_copy_model_data(g, am, tm, blob.data(), blob.size(), infos.data(), (uint32_t)infos.size(), &total);
    if (r) return r;
    for (size_t i = 0; i < infos.size(); i++) {
        const ArtGlbCopyInfo &ci = infos[i];
        if (ci.single_mesh_element_size != 48) return gfail(ART_E_INVALID, "art_scene_add_glb: vertices are not the 48-byte interleave");
        if (ci.image_format != F_R8G8B8A8 && ci.image_format != F_B8G8R8A8) return gfail(ART_E_INVALID, "art_scene_add_glb: textures must be coerced to RGBA8/BGRA8 (open with coerce_format 1 or 2)");
        if (ci.image_layers != 3) return gfail(ART_E_INVALID, "art_scene_add_glb: expected albedo + ORM + normal layers");
        std::vector<uint8_t> rgba(blob.begin() + ci.image_buffer_offset, blob.begin() + ci.image_buffer_offset + ci.image_size);
        if (ci.image_format == F_B8G8R8A8) for (size_t t = 0; t + 3 < rgba.size(); t += 4) std::swap(rgba[t], rgba[t + 2]); // the sampler returns logical RGBA
        uint32_t id = 0;
        r = art_scene_add_primitive(ctx, (const ArtVertex *)(blob.data() + ci.mesh_buffer_offset), (uint32_t)(ci.mesh_size / 48), blob.data() + ci.indices_buffer_offset,
                                    (uint32_t)(ci.indices_size / ci.single_index_size), ci.single_index_size, rgba.data(), ci.image_width, ci.image_height, model3x4, &id);
        if (r) { g_glb_err = art_last_error(); return r; }
        if (i == 0 && first_primitive_id) *first_primitive_id = id;
    }
    if (n_primitives) *n_primitives = (uint32_t)infos.size();
    return ART_OK;
}
int32_t art_scene_add_glb(ArtContext *ctx, ArtGlb *g, const float model3x4[12], uint32_t *first_primitive_id, uint32_t *n_primitives) {
    return glb_guard("art_scene_add_glb", [&] { return glb_add_impl(ctx, g, model3x4, first_primitive_id, n_primitives); });
}

} // extern "C"
