// Wavefront OBJ loader — behavioural restatement of reference src/loaders/obj.rs:13-107.
// Grammar: tokens are separated by single spaces (`line.split(' ')`, obj.rs:24);
// `v x y z`, `vt u v [w]`, `vn x y z` (normalised on load, obj.rs:48), `f a/b/c ...` or
// `f a//c ...` with 1-based or negative indices; every other line is ignored.
// Where the reference panics (unwrap on a bad number, >3 vertices per face) this returns an
// error string instead.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "host_internal.h"

namespace rth {

static bool parse_f64_tok(const char* b, const char* e, double* out) {
    if (b == e) return false;
    // strtod needs NUL termination; tokens are short, copy to a stack buffer.
    char buf[64];
    size_t n = size_t(e - b);
    if (n >= sizeof buf) return false;
    if (*b == ' ' || *b == '\t') return false;
    std::memcpy(buf, b, n);
    buf[n] = 0;
    char* end = nullptr;
    *out = std::strtod(buf, &end);
    return end == buf + n;
}

static bool parse_i32_tok(const char* b, const char* e, long* out) {
    if (b == e) return false;
    char buf[32];
    size_t n = size_t(e - b);
    if (n >= sizeof buf) return false;
    std::memcpy(buf, b, n);
    buf[n] = 0;
    char* end = nullptr;
    *out = std::strtol(buf, &end, 10);
    return end == buf + n;
}

std::unique_ptr<MeshData> load_obj(const std::string& path, std::string* log, std::string* err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        *err = "cannot open " + path;
        return nullptr;
    }
    std::fseek(f, 0, SEEK_END);
    long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::string text;
    text.resize(size_t(size));
    if (size > 0 && std::fread(&text[0], 1, size_t(size), f) != size_t(size)) {
        std::fclose(f);
        *err = "short read on " + path;
        return nullptr;
    }
    std::fclose(f);

    auto mesh = std::make_unique<MeshData>();
    const char* p = text.data();
    const char* end = p + text.size();
    size_t line_no = 0;
    std::vector<std::pair<const char*, const char*>> toks;
    while (p < end) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
        const char* le = nl ? nl : end;
        const char* next = nl ? nl + 1 : end;
        if (le > p && le[-1] == '\r') le--;  // BufRead::lines strips CRLF
        line_no++;
        // split(' ')
        toks.clear();
        const char* tb = p;
        for (const char* c = p;; c++) {
            if (c == le || *c == ' ') {
                toks.emplace_back(tb, c);
                tb = c + 1;
                if (c == le) break;
            }
        }
        p = next;
        size_t cl = size_t(toks[0].second - toks[0].first);
        const char* cmd = toks[0].first;
        auto fail = [&](const char* what) {
            *err = path + ":" + std::to_string(line_no) + ": " + what;
            return nullptr;
        };
        if (cl == 1 && cmd[0] == 'v') {
            double v[3];
            if (toks.size() < 4) return fail("vertex needs 3 components");
            for (size_t i = 1; i < toks.size(); i++) {
                double x;
                if (!parse_f64_tok(toks[i].first, toks[i].second, &x)) return fail("bad number");
                if (i <= 3) v[i - 1] = x;
            }
            mesh->positions.insert(mesh->positions.end(), v, v + 3);
        } else if (cl == 2 && cmd[0] == 'v' && cmd[1] == 't') {
            double v[3] = {0, 0, 0};
            if (toks.size() < 3) return fail("uv needs 2 components");
            for (size_t i = 1; i < toks.size(); i++) {
                double x;
                if (!parse_f64_tok(toks[i].first, toks[i].second, &x)) return fail("bad number");
                if (i <= 3) v[i - 1] = x;
            }
            mesh->uvs.insert(mesh->uvs.end(), v, v + 3);
        } else if (cl == 2 && cmd[0] == 'v' && cmd[1] == 'n') {
            double v[3];
            if (toks.size() < 4) return fail("normal needs 3 components");
            for (size_t i = 1; i < toks.size(); i++) {
                double x;
                if (!parse_f64_tok(toks[i].first, toks[i].second, &x)) return fail("bad number");
                if (i <= 3) v[i - 1] = x;
            }
            double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            v[0] /= len; v[1] /= len; v[2] /= len;  // Vec4::to_unit = self / length
            mesh->normals.insert(mesh->normals.end(), v, v + 3);
        } else if (cl == 1 && cmd[0] == 'f') {
            uint32_t iv[3] = {0, 0, 0}, in[3] = {0, 0, 0};
            int32_t iu[3] = {-1, -1, -1};
            if (toks.size() > 4) return fail("only triangles are supported");
            for (size_t i = 1; i < toks.size(); i++) {
                // split('/')
                const char* parts_b[3];
                const char* parts_e[3];
                int np = 0;
                const char* pb = toks[i].first;
                for (const char* c = toks[i].first;; c++) {
                    if (c == toks[i].second || *c == '/') {
                        if (np < 3) { parts_b[np] = pb; parts_e[np] = c; }
                        np++;
                        pb = c + 1;
                        if (c == toks[i].second) break;
                    }
                }
                if (np < 3) return fail("face vertex must be v/vt/vn or v//vn");
                long a, b = 0, c2;
                if (!parse_i32_tok(parts_b[0], parts_e[0], &a)) return fail("bad face index");
                bool has_uv = parts_b[1] != parts_e[1];
                if (has_uv && !parse_i32_tok(parts_b[1], parts_e[1], &b)) return fail("bad face index");
                if (!parse_i32_tok(parts_b[2], parts_e[2], &c2)) return fail("bad face index");
                long nv = long(mesh->positions.size() / 3), nn = long(mesh->normals.size() / 3),
                     nu = long(mesh->uvs.size() / 3);
                iv[i - 1] = uint32_t(a > 0 ? a - 1 : a < 0 ? nv + a : 0);  // obj.rs:63-67
                if (has_uv) iu[i - 1] = int32_t(b > 0 ? b - 1 : b < 0 ? nu + b : 0);
                in[i - 1] = uint32_t(c2 > 0 ? c2 - 1 : c2 < 0 ? nn + c2 : 0);
            }
            bool has_uvs = iu[0] >= 0 && iu[1] >= 0 && iu[2] >= 0;  // obj.rs:81
            mesh->tri_pos.insert(mesh->tri_pos.end(), iv, iv + 3);
            mesh->tri_nrm.insert(mesh->tri_nrm.end(), in, in + 3);
            for (int k = 0; k < 3; k++) mesh->tri_uv.push_back(has_uvs ? iu[k] : -1);
            if (has_uvs) mesh->any_uv = true;
        }
    }
    size_t ntri = mesh->tri_pos.size() / 3;
    // Index validation (the reference would panic on first use of a bad index).
    size_t nv = mesh->positions.size() / 3, nn = mesh->normals.size() / 3, nu = mesh->uvs.size() / 3;
    for (size_t i = 0; i < ntri * 3; i++) {
        if (mesh->tri_pos[i] >= nv || mesh->tri_nrm[i] >= nn ||
            (mesh->tri_uv[i] >= 0 && size_t(mesh->tri_uv[i]) >= nu)) {
            *err = path + ": face index out of range";
            return nullptr;
        }
    }
    *log += "Loaded " + std::to_string(ntri) + " tris\n";  // obj.rs:99
    return mesh;
}

}  // namespace rth
