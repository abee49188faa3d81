// gen_dragon — deterministic stand-in for the reference's `scenes/resource/dragon_high.obj`.
//
// The reference's cornell_dragon scene (scenes/cornell_dragon:30) loads a ~870k-triangle scan
// that is NOT in the checkout (/root/reference/.MISSING_LARGE_BLOBS).  This tool writes a
// closed, curved, high-frequency surface with the same triangle budget so that traversal
// statistics are comparable: a (2,3) torus-knot tube, 660 x 660 quads = 871 200 triangles,
// radially displaced by a sum of sines ("scales"), standing on y = 0 and fitting
// x,z in [-3,3], y in [0,6.5] so that the scene's `s=60 ry=225 t=267.5,0.5,277.5` puts it on
// the floor inside the 555-unit box.  No RNG: the output is a pure function of (NU, NV).
// Grammar = what src/loaders/obj.rs accepts: `v x y z`, `vn x y z`, `f a//a b//b c//c`.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct P3 {
    double x, y, z;
};
static P3 sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static P3 add(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static P3 mul(P3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static P3 unit(P3 a) {
    double l = std::sqrt(dot(a, a));
    return {a.x / l, a.y / l, a.z / l};
}

static const double kPi = 3.14159265358979323846;

static P3 knot(double t) {  // (2,3) torus knot, major radius 1.45, minor 0.62, upright (axis = z -> lies in the XY plane)
    double r = 1.45 + 0.62 * std::cos(3.0 * t);
    return {r * std::cos(2.0 * t), r * std::sin(2.0 * t), 0.62 * std::sin(3.0 * t) * 1.6};
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "scenes/resource/dragon_high.obj";
    int NU = argc > 2 ? std::atoi(argv[2]) : 660;  // along the knot
    int NV = argc > 3 ? std::atoi(argv[3]) : 660;  // around the tube
    if (NU < 3 || NV < 3) return 2;

    std::vector<P3> pos(size_t(NU) * NV);
    // Rotation-minimising-ish frame: project a fixed up vector, fall back when nearly parallel.
    for (int i = 0; i < NU; i++) {
        double t = 2.0 * kPi * double(i) / double(NU);
        double h = 1e-4;
        P3 c = knot(t);
        P3 tan = unit(sub(knot(t + h), knot(t - h)));
        P3 up = {0.0, 0.0, 1.0};
        P3 n = sub(up, mul(tan, dot(up, tan)));
        if (dot(n, n) < 1e-6) {
            up = {1.0, 0.0, 0.0};
            n = sub(up, mul(tan, dot(up, tan)));
        }
        n = unit(n);
        P3 b = cross(tan, n);
        for (int j = 0; j < NV; j++) {
            double a = 2.0 * kPi * double(j) / double(NV);
            // tube radius with "scales": three incommensurate ripples
            double rad = 0.36 + 0.030 * std::sin(36.0 * t + 5.0 * a) + 0.018 * std::sin(90.0 * t - 11.0 * a) +
                         0.010 * std::sin(210.0 * t + 23.0 * a) + 0.05 * std::sin(3.0 * t) * std::cos(2.0 * a);
            P3 d = add(mul(n, std::cos(a)), mul(b, std::sin(a)));
            pos[size_t(i) * NV + j] = add(c, mul(d, rad));
        }
    }
    // Stand it on y = 0 and fit the target box.
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (auto& p : pos) {
        double v[3] = {p.x, p.y, p.z};
        for (int k = 0; k < 3; k++) {
            if (v[k] < lo[k]) lo[k] = v[k];
            if (v[k] > hi[k]) hi[k] = v[k];
        }
    }
    double sx = 6.0 / (hi[0] - lo[0]), sy = 6.5 / (hi[1] - lo[1]), sz = 6.0 / (hi[2] - lo[2]);
    double s = std::fmin(sx, std::fmin(sy, sz));
    double cx = 0.5 * (lo[0] + hi[0]), cz = 0.5 * (lo[2] + hi[2]);
    for (auto& p : pos) {
        p.x = (p.x - cx) * s;
        p.y = (p.y - lo[1]) * s;
        p.z = (p.z - cz) * s;
    }
    // Smooth vertex normals from the parametric neighbours.
    std::vector<P3> nrm(pos.size());
    for (int i = 0; i < NU; i++)
        for (int j = 0; j < NV; j++) {
            auto at = [&](int ii, int jj) { return pos[size_t((ii + NU) % NU) * NV + size_t((jj + NV) % NV)]; };
            P3 du = sub(at(i + 1, j), at(i - 1, j));
            P3 dv = sub(at(i, j + 1), at(i, j - 1));
            nrm[size_t(i) * NV + j] = unit(cross(dv, du));
        }
    // Orientation: make normals point away from the tube centre line.
    {
        P3 c0 = {0, 0, 0};
        for (int j = 0; j < NV; j++) c0 = add(c0, pos[j]);
        c0 = mul(c0, 1.0 / NV);
        if (dot(nrm[0], sub(pos[0], c0)) < 0)
            for (auto& n : nrm) n = mul(n, -1.0);
    }

    FILE* f = std::fopen(path, "wb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path);
        return 1;
    }
    std::vector<char> buf(1 << 22);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    std::fprintf(f, "# deterministic stand-in for dragon_high.obj: (2,3) torus-knot tube %dx%d quads, %d triangles\n", NU, NV, 2 * NU * NV);
    std::fprintf(f, "o DragonStandIn\n");
    for (auto& p : pos) std::fprintf(f, "v %.9g %.9g %.9g\n", p.x, p.y, p.z);
    for (auto& n : nrm) std::fprintf(f, "vn %.9g %.9g %.9g\n", n.x, n.y, n.z);
    // Triangle winding must be counter-clockwise seen from outside (the reference culls back
    // faces: det < EPSILON, src/object/mesh.rs:76-80): check it on the first quad and flip if needed.
    auto idx = [&](int i, int j) { return size_t((i + NU) % NU) * NV + size_t((j + NV) % NV); };
    bool flip;
    {
        P3 a = pos[idx(0, 0)], b = pos[idx(1, 0)], c = pos[idx(1, 1)];
        P3 g = cross(sub(b, a), sub(c, a));
        flip = dot(g, nrm[idx(0, 0)]) < 0;
    }
    for (int i = 0; i < NU; i++)
        for (int j = 0; j < NV; j++) {
            size_t a = idx(i, j) + 1, b = idx(i + 1, j) + 1, c = idx(i + 1, j + 1) + 1, d = idx(i, j + 1) + 1;
            if (flip) {
                std::fprintf(f, "f %zu//%zu %zu//%zu %zu//%zu\n", a, a, c, c, b, b);
                std::fprintf(f, "f %zu//%zu %zu//%zu %zu//%zu\n", a, a, d, d, c, c);
            } else {
                std::fprintf(f, "f %zu//%zu %zu//%zu %zu//%zu\n", a, a, b, b, c, c);
                std::fprintf(f, "f %zu//%zu %zu//%zu %zu//%zu\n", a, a, c, c, d, d);
            }
        }
    std::fclose(f);
    return 0;
}
