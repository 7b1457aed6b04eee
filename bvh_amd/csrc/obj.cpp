// obj.cpp — Wavefront OBJ ingest for the scenes of the reference's harness (host code, no GPU).
//
// Replaces `obj::load_obj::<Triangle>` as the reference uses it (src/testbase.rs:445-487, 619-634;
// crate obj-rs 0.7, un-vendored): parse `v` / `f` statements, then `FromRawVertex::process` turns every
// polygon into a triangle FAN — anchor = first vertex, (anchor, second, third), second = third
// (testbase.rs:461-469) — using positions only for all four polygon kinds P, PT, PN, PTN (:473-482).
// A polygon with two vertices yields nothing; one with a single vertex is an error (the reference's
// `unwrap()` panics).  Indices are 1-based, negative ones count back from the vertices read so far.
// Output: n x 9 floats [a xyz, b xyz, c xyz] = the fields of testbase.rs Triangle (:316-323), ready for
// bvhgpu_tree_set_triangles_f32; the AABBs are Triangle::new's empty.grow(a).grow(b).grow(c) (:325-333).
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/bvh_mi355x.h"

namespace {

thread_local std::string g_obj_err;

struct Cursor {
    const char* p;
    const char* end;
    size_t line = 1;
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// advance over blanks; a backslash-newline continues the statement
inline void skip_blanks(Cursor& c) {
    while (c.p < c.end) {
        if (is_space(*c.p)) { c.p++; continue; }
        if (*c.p == '\\' && c.p + 1 < c.end && (c.p[1] == '\n' || (c.p[1] == '\r' && c.p + 2 < c.end && c.p[2] == '\n'))) {
            c.p += (c.p[1] == '\n') ? 2 : 3;
            c.line++;
            continue;
        }
        break;
    }
}
inline bool at_eol(const Cursor& c) { return c.p >= c.end || *c.p == '\n' || *c.p == '#'; }
inline void skip_line(Cursor& c) {
    while (c.p < c.end && *c.p != '\n') c.p++;
    if (c.p < c.end) { c.p++; c.line++; }
}

bool parse_float(Cursor& c, float* out) {
    skip_blanks(c);
    if (at_eol(c)) return false;
    char buf[64];
    size_t n = 0;
    while (c.p < c.end && !is_space(*c.p) && *c.p != '\n' && *c.p != '#' && n + 1 < sizeof buf) buf[n++] = *c.p++;
    buf[n] = 0;
    // Rust's f32::from_str grammar: decimal digits, sign, '.', exponent, "inf"/"infinity"/"nan" — no hex floats
    for (size_t i = 0; i < n; i++) {
        const char ch = buf[i];
        const bool ok = (ch >= '0' && ch <= '9') || ch == '+' || ch == '-' || ch == '.' || ch == 'e' || ch == 'E' ||
                        ch == 'i' || ch == 'n' || ch == 'f' || ch == 'a' || ch == 't' || ch == 'y' || ch == 'I' || ch == 'N';
        if (!ok) return false;
    }
    char* endp = nullptr;
    errno = 0;
    const float v = strtof(buf, &endp);   // correctly rounded, like Rust
    if (endp == buf || *endp != 0) return false;
    *out = v;
    return true;
}

// a decimal integer with an optional sign, read inside [c.p, c.end) only — the buffer need not be NUL-terminated, and
// unlike strtoll no white space (in particular no newline) is skipped in front of it
bool parse_int(Cursor& c, long long* out) {
    const char* p = c.p;
    bool neg = false;
    if (p < c.end && (*p == '+' || *p == '-')) { neg = *p == '-'; p++; }
    if (p >= c.end || *p < '0' || *p > '9') return false;
    long long v = 0;
    while (p < c.end && *p >= '0' && *p <= '9') {
        if (v > (0x7FFFFFFFFFFFFFFFll - 9) / 10) return false;   // overflow: not an index any OBJ file can mean
        v = v * 10 + (*p - '0');
        p++;
    }
    c.p = p;
    *out = neg ? -v : v;
    return true;
}

// one face vertex "p", "p/t", "p//n" or "p/t/n": returns the position index and the kind (bit0: t, bit1: n)
bool parse_face_vertex(Cursor& c, long long* pos, int* kind) {
    skip_blanks(c);
    if (at_eol(c)) return false;
    long long v = 0, ignored = 0;
    if (!parse_int(c, &v)) return false;
    *pos = v;
    int k = 0;
    if (c.p < c.end && *c.p == '/') {
        c.p++;
        if (c.p < c.end && *c.p != '/' && !is_space(*c.p) && *c.p != '\n') {
            if (!parse_int(c, &ignored)) return false;
            k |= 1;
        }
        if (c.p < c.end && *c.p == '/') {
            c.p++;
            if (!parse_int(c, &ignored)) return false;
            k |= 2;
        }
    }
    if (c.p < c.end && !is_space(*c.p) && *c.p != '\n' && *c.p != '#' && *c.p != '\\') return false;
    *kind = k;
    return true;
}

int obj_fail(const Cursor& c, const char* what) {
    g_obj_err = std::string("OBJ line ") + std::to_string(c.line) + ": " + what;
    return BVHGPU_INVALID_ARG;
}

}  // namespace

extern "C" {

const char* bvhgpu_obj_last_error(void) { return g_obj_err.c_str(); }

int bvhgpu_obj_parse(const char* text, size_t len, float** tris_out, size_t* n_tris_out, float bounds_out[6]) {
    if (!tris_out || !n_tris_out || (len && !text)) return BVHGPU_INVALID_ARG;
    *tris_out = nullptr;
    *n_tris_out = 0;
    try {
        std::vector<float> pts;    // x y z per vertex (w is parsed and dropped, testbase.rs:455-458)
        std::vector<float> tris;
        std::vector<long long> face;
        Cursor c{text, text + len};
        while (c.p < c.end) {
            skip_blanks(c);
            if (c.p >= c.end) break;
            if (*c.p == '\n') { c.p++; c.line++; continue; }
            if (*c.p == '#') { skip_line(c); continue; }
            const char* kw = c.p;
            while (c.p < c.end && !is_space(*c.p) && *c.p != '\n' && *c.p != '#') c.p++;
            const size_t kl = (size_t)(c.p - kw);
            auto is = [&](const char* s) { return kl == strlen(s) && memcmp(kw, s, kl) == 0; };
            if (is("v")) {
                float v[4] = {0, 0, 0, 1};
                int got = 0;
                while (got < 4 && parse_float(c, &v[got])) got++;
                skip_blanks(c);
                if (got < 3 || !at_eol(c)) return obj_fail(c, "a vertex needs 3 or 4 numbers");
                pts.push_back(v[0]); pts.push_back(v[1]); pts.push_back(v[2]);
            } else if (is("f")) {
                face.clear();
                int kind0 = -1;
                long long pi;
                int kind;
                while (true) {
                    skip_blanks(c);
                    if (at_eol(c)) break;
                    if (!parse_face_vertex(c, &pi, &kind)) return obj_fail(c, "malformed face vertex");
                    if (kind0 < 0) kind0 = kind;
                    else if (kind != kind0) return obj_fail(c, "face mixes vertex formats");
                    const long long nv = (long long)(pts.size() / 3);
                    long long idx = pi > 0 ? pi - 1 : (pi < 0 ? nv + pi : -1);
                    if (idx < 0 || idx >= nv) return obj_fail(c, "face index out of range");
                    face.push_back(idx);
                }
                if (face.size() < 2) return obj_fail(c, "a face needs at least two vertices");
                // triangle fan (testbase.rs:461-469)
                const float* anchor = &pts[3 * (size_t)face[0]];
                const float* second = &pts[3 * (size_t)face[1]];
                for (size_t k = 2; k < face.size(); k++) {
                    const float* third = &pts[3 * (size_t)face[k]];
                    tris.insert(tris.end(), anchor, anchor + 3);
                    tris.insert(tris.end(), second, second + 3);
                    tris.insert(tris.end(), third, third + 3);
                    second = third;
                }
            } else if (is("vt") || is("vn") || is("vp") || is("g") || is("o") || is("s") || is("usemtl") || is("mtllib") ||
                       is("l") || is("p")) {
                // parsed by obj-rs, irrelevant for Triangle (process ignores tex coords and normals, :448-449)
            } else {
                return obj_fail(c, "unexpected statement");
            }
            skip_line(c);
        }
        const size_t n = tris.size() / 9;
        if (bounds_out) {   // load_sponza_scene: bounds = join of the triangle AABBs (testbase.rs:628-631)
            float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (size_t i = 0; i < tris.size(); i += 3)
                for (int k = 0; k < 3; k++) {
                    const float x = tris[i + k];   // join orders -0 < +0 (common.hpp tmin / tmax)
                    b[k] = (x < b[k] || (x == b[k] && std::signbit(x))) ? x : b[k];
                    b[3 + k] = (x > b[3 + k] || (x == b[3 + k] && !std::signbit(x))) ? x : b[3 + k];
                }
            memcpy(bounds_out, b, sizeof b);
        }
        if (n) {
            float* out = static_cast<float*>(malloc(tris.size() * sizeof(float)));
            if (!out) { g_obj_err = "out of memory"; return BVHGPU_OOM; }
            memcpy(out, tris.data(), tris.size() * sizeof(float));
            *tris_out = out;
        }
        *n_tris_out = n;
        return BVHGPU_OK;
    } catch (const std::bad_alloc&) {
        g_obj_err = "out of memory";
        return BVHGPU_OOM;
    }
}

void bvhgpu_obj_free(float* tris) { free(tris); }

// Triangle::new's aabb (testbase.rs:325-333): empty.grow(a).grow(b).grow(c) → n x [min xyz, max xyz]
int bvhgpu_triangles_aabbs_f32(const float* tris, size_t n, float* aabbs_out) {
    if (n && (!tris || !aabbs_out)) return BVHGPU_INVALID_ARG;
    for (size_t i = 0; i < n; i++) {
        const float* t = tris + 9 * i;
        float* o = aabbs_out + 6 * i;
        for (int k = 0; k < 3; k++) {
            float mn = INFINITY, mx = -INFINITY;
            for (int v = 0; v < 3; v++) {
                const float x = t[3 * v + k];
                mn = (x < mn || (x == mn && std::signbit(x))) ? x : mn;
                mx = (x > mx || (x == mx && !std::signbit(x))) ? x : mx;
            }
            o[k] = mn; o[3 + k] = mx;
        }
    }
    return BVHGPU_OK;
}

}  // extern "C"
