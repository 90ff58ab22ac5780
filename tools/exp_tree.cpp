// exp_tree.cpp -- CPU experiment: how much would a better BINARY tree under the 8-wide collapse be worth?
// Builds binary trees over the bench scene's triangles with different split searches (upstream's 8 bins on the centroid
// range, more bins, a full sweep, sweep + spatial splits) -- always split down to single triangles, as upstream does -- and
// prices each by the collapse plan of walk_tree.hip (k8_plan's dynamic program: leaf slots of <= 3 triangles bounded by
// their own triangles, a node step priced at RT_W8_DP_CNODE triangle tests), i.e. the expected node steps and triangle
// tests of a long random ray.
//   python - <<< 'from raytracer_amd import scenes; scenes.interior("bathroom").verts[:, :, :3].tofile("/tmp/tris.bin")'
//   g++ -O2 -std=c++17 tools/exp_tree.cpp -o /tmp/exp_tree && /tmp/exp_tree /tmp/tris.bin
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

struct Box {
    float lo[3], hi[3];
    Box() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    bool valid() const { return lo[0] <= hi[0]; }
    double area() const {
        if (!valid()) return 0.0;
        double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return 2.0 * (x * y + y * z + x * z);
    }
};
// a reference: a triangle and the box of the part of it that this reference stands for (spatial splits clip it)
struct Ref { uint32_t tri; Box b; };
struct Node { Box box; int left = -1, right = -1; uint32_t count = 0; };

static const float* V; static uint32_t NT;
static std::vector<Node> nodes;
static int MODE, BINS; static double SPATIAL_ALPHA = 1e-5; static double ROOT_AREA;
static uint64_t n_spatial = 0, n_refs_final = 0;

static Box tri_box(uint32_t t) { Box b; for (int k = 0; k < 3; k++) b.grow(V + (size_t)t * 9 + k * 3); return b; }
static inline float centre(const Box& b, int a) { return 0.5f * b.lo[a] + 0.5f * b.hi[a]; }

// clip the triangle t to the slab lo <= x_a <= hi, return the box of the clipped polygon intersected with `within`
static Box clip_box(uint32_t t, int a, float lo, float hi, const Box& within) {
    double poly[2][16][3]; int n = 3, cur = 0;
    for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) poly[0][k][c] = V[(size_t)t * 9 + k * 3 + c];
    for (int side = 0; side < 2; side++) {
        const double pl = side ? hi : lo; const double sg = side ? -1.0 : 1.0;
        int m = 0;
        for (int i = 0; i < n; i++) {
            const double* p = poly[cur][i]; const double* q = poly[cur][(i + 1) % n];
            const double dp = sg * (p[a] - pl), dq = sg * (q[a] - pl);
            if (dp >= 0) { memcpy(poly[cur ^ 1][m++], p, sizeof(double) * 3); }
            if ((dp > 0 && dq < 0) || (dp < 0 && dq > 0)) {
                const double w = dp / (dp - dq);
                for (int c = 0; c < 3; c++) poly[cur ^ 1][m][c] = p[c] + (q[c] - p[c]) * w;
                poly[cur ^ 1][m][a] = pl; m++;
            }
        }
        cur ^= 1; n = m;
        if (!n) break;
    }
    Box b;
    for (int i = 0; i < n; i++) { float p[3] = { (float)poly[cur][i][0], (float)poly[cur][i][1], (float)poly[cur][i][2] }; b.grow(p); }
    if (b.valid()) for (int c = 0; c < 3; c++) { b.lo[c] = std::max(b.lo[c], within.lo[c]); b.hi[c] = std::min(b.hi[c], within.hi[c]); }
    return b;
}

static int build(std::vector<Ref>& refs) {
    const int ni = (int)nodes.size(); nodes.emplace_back();
    Box box; for (auto& r : refs) box.grow(r.b);
    nodes[ni].box = box; nodes[ni].count = (uint32_t)refs.size();
    const size_t n = refs.size();
    if (n <= 1) { n_refs_final += n; return ni; }
    const double pa = box.area();
    // object split
    double best = DBL_MAX; int baxis = -1; float bpos = 0; size_t bsplit = 0; Box obl, obr;
    std::vector<uint32_t> order(n);
    for (int a = 0; a < 3; a++) {
        float cmin = FLT_MAX, cmax = -FLT_MAX;
        for (auto& r : refs) { float c = centre(r.b, a); cmin = std::min(cmin, c); cmax = std::max(cmax, c); }
        if (cmin == cmax) continue;
        if (MODE == 0) {   // bins on the centroid range
            std::vector<Box> bb(BINS); std::vector<int> bc(BINS, 0);
            const float scale = BINS / (cmax - cmin);
            for (auto& r : refs) { int b = std::min(BINS - 1, (int)((centre(r.b, a) - cmin) * scale)); bc[b]++; bb[b].grow(r.b); }
            std::vector<double> ra(BINS); std::vector<int> rc(BINS);
            Box rb; int rs = 0;
            for (int i = BINS - 1; i >= 1; i--) { rs += bc[i]; rb.grow(bb[i]); ra[i] = rb.area(); rc[i] = rs; }
            Box lb; int ls = 0;
            for (int i = 0; i < BINS - 1; i++) {
                ls += bc[i]; lb.grow(bb[i]);
                if (!ls || !rc[i + 1]) continue;
                const double cost = lb.area() * ls + ra[i + 1] * rc[i + 1];
                if (cost < best) { best = cost; baxis = a; bpos = cmin + (cmax - cmin) / BINS * (i + 1); }
            }
        } else {           // full sweep over the sorted centroids
            std::iota(order.begin(), order.end(), 0u);
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return centre(refs[x].b, a) < centre(refs[y].b, a); });
            std::vector<double> ra(n);
            Box rb;
            for (size_t i = n; i-- > 1;) { rb.grow(refs[order[i]].b); ra[i] = rb.area(); }
            Box lb;
            for (size_t i = 0; i + 1 < n; i++) {
                lb.grow(refs[order[i]].b);
                const double cost = lb.area() * (double)(i + 1) + ra[i + 1] * (double)(n - i - 1);
                if (cost < best) { best = cost; baxis = a; bsplit = i + 1; }
            }
        }
    }
    // spatial split (Stich et al. 2009), only where the object split's children overlap noticeably
    bool spatial = false; int saxis = -1; float spos = 0; double sbest = DBL_MAX;
    if (MODE == 2 && baxis >= 0) {
        // overlap of the best object split's children
        std::vector<Ref> tl, tr;
        Box lb, rb;
        {
            std::iota(order.begin(), order.end(), 0u);
            const int a = baxis;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return centre(refs[x].b, a) < centre(refs[y].b, a); });
            for (size_t i = 0; i < n; i++) (i < bsplit ? lb : rb).grow(refs[order[i]].b);
        }
        Box ov; bool has = true;
        for (int c = 0; c < 3; c++) { ov.lo[c] = std::max(lb.lo[c], rb.lo[c]); ov.hi[c] = std::min(lb.hi[c], rb.hi[c]); if (ov.lo[c] > ov.hi[c]) has = false; }
        if (has && ov.area() / ROOT_AREA > SPATIAL_ALPHA) {
            const int SB = 32;
            for (int a = 0; a < 3; a++) {
                const float lo = box.lo[a], hi = box.hi[a];
                if (!(hi > lo)) continue;
                const float w = (hi - lo) / SB;
                std::vector<Box> bb(SB); std::vector<int> en(SB, 0), ex(SB, 0);
                for (auto& r : refs) {
                    int b0 = std::min(SB - 1, std::max(0, (int)((r.b.lo[a] - lo) / w))), b1 = std::min(SB - 1, std::max(0, (int)((r.b.hi[a] - lo) / w)));
                    en[b0]++; ex[b1]++;
                    if (b0 == b1) { bb[b0].grow(r.b); continue; }
                    for (int b = b0; b <= b1; b++) {
                        Box cb = clip_box(r.tri, a, lo + w * b, b == SB - 1 ? hi : lo + w * (b + 1), r.b);
                        if (cb.valid()) bb[b].grow(cb);
                    }
                }
                std::vector<double> ra(SB); std::vector<int> rc(SB);
                Box rb2; int rs = 0;
                for (int i = SB - 1; i >= 1; i--) { rs += ex[i]; rb2.grow(bb[i]); ra[i] = rb2.area(); rc[i] = rs; }
                Box lb2; int ls = 0;
                for (int i = 0; i < SB - 1; i++) {
                    ls += en[i]; lb2.grow(bb[i]);
                    if (!ls || !rc[i + 1]) continue;
                    const double cost = lb2.area() * ls + ra[i + 1] * rc[i + 1];
                    if (cost < sbest) { sbest = cost; saxis = a; spos = lo + w * (i + 1); }
                }
            }
            if (sbest < best * 0.999) spatial = true;
        }
    }
    std::vector<Ref> L, R;
    if (spatial) {
        for (auto& r : refs) {
            if (r.b.hi[saxis] <= spos) L.push_back(r);
            else if (r.b.lo[saxis] >= spos) R.push_back(r);
            else {
                Ref a = r, b = r;
                a.b = clip_box(r.tri, saxis, box.lo[saxis], spos, r.b); b.b = clip_box(r.tri, saxis, spos, box.hi[saxis], r.b);
                if (a.b.valid()) L.push_back(a);
                if (b.b.valid()) R.push_back(b);
                if (!a.b.valid() && !b.b.valid()) L.push_back(r);
            }
        }
        if (L.empty() || R.empty() || L.size() == n + 0 && R.size() == n + 0) { spatial = false; L.clear(); R.clear(); }
        else n_spatial++;
    }
    if (!spatial) {
        if (baxis < 0) { n_refs_final += n; return ni; }   // all centroids coincide: a leaf
        if (MODE == 0) { for (auto& r : refs) (centre(r.b, baxis) < bpos ? L : R).push_back(r); }
        else {
            std::iota(order.begin(), order.end(), 0u);
            const int a = baxis;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return centre(refs[x].b, a) < centre(refs[y].b, a); });
            for (size_t i = 0; i < n; i++) (i < bsplit ? L : R).push_back(refs[order[i]]);
        }
        if (L.empty() || R.empty()) { n_refs_final += n; return ni; }
    }
    (void)pa;
    std::vector<Ref>().swap(refs);
    const int l = build(L); std::vector<Ref>().swap(L);
    const int r = build(R);
    nodes[ni].left = l; nodes[ni].right = r;
    return ni;
}

// k8_plan's dynamic program
struct Cost { double c[8], nn[8], nt[8]; };
static double CNODE = 1.5; static uint32_t LEAF_MAX = 3;
static Cost plan(int ni) {
    const Node& nd = nodes[ni];
    Cost o;
    const double A = nd.box.area() / ROOT_AREA;
    if (nd.left < 0) {
        double c = A * nd.count, nn = 0, nt = c;
        if (nd.count > LEAF_MAX) { nn = A; nt = c * 0.5; c = A * CNODE + c * 0.5; }
        for (int k = 1; k < 8; k++) { o.c[k] = c; o.nn[k] = nn; o.nt[k] = nt; }
        return o;
    }
    const Cost cl = plan(nd.left), cr = plan(nd.right);
    double D[9], Dn[9], Dt[9];
    for (int j = 2; j <= 8; j++) {
        D[j] = DBL_MAX;
        for (int k = std::max(1, j - 7); k <= 7 && k < j; k++) {
            const double c = cl.c[k] + cr.c[j - k];
            if (c < D[j]) { D[j] = c; Dn[j] = cl.nn[k] + cr.nn[j - k]; Dt[j] = cl.nt[k] + cr.nt[j - k]; }
        }
    }
    if (nd.count <= LEAF_MAX) { o.c[1] = A * nd.count; o.nn[1] = 0; o.nt[1] = A * nd.count; }
    else { o.c[1] = A * CNODE + D[8]; o.nn[1] = A + Dn[8]; o.nt[1] = Dt[8]; }
    for (int k = 2; k <= 7; k++) {
        if (D[k] < o.c[k - 1]) { o.c[k] = D[k]; o.nn[k] = Dn[k]; o.nt[k] = Dt[k]; } else { o.c[k] = o.c[k - 1]; o.nn[k] = o.nn[k - 1]; o.nt[k] = o.nt[k - 1]; }
    }
    return o;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    NT = (uint32_t)(sz / 36); std::vector<float> v((size_t)NT * 9);
    if (fread(v.data(), 36, NT, f) != NT) return 1;
    fclose(f); V = v.data();
    printf("%u triangles\n", NT);
    struct Cfg { const char* name; int mode, bins; } cfgs[] = { { "8 bins (upstream's search)", 0, 8 }, { "32 bins", 0, 32 }, { "128 bins", 0, 128 }, { "full sweep", 1, 0 }, { "sweep + spatial splits", 2, 0 } };
    for (auto& c : cfgs) {
        if (argc > 2 && atoi(argv[2]) != (int)(&c - cfgs)) continue;
        MODE = c.mode; BINS = c.bins; nodes.clear(); n_spatial = 0; n_refs_final = 0;
        std::vector<Ref> refs(NT);
        Box root;
        for (uint32_t t = 0; t < NT; t++) { refs[t].tri = t; refs[t].b = tri_box(t); root.grow(refs[t].b); }
        ROOT_AREA = root.area();
        nodes.reserve((size_t)NT * 3);
        build(refs);
        const Cost o = plan(0);
        // binary SAH for reference: sum of inner areas + leaf areas * count
        double sah_n = 0, sah_t = 0;
        for (auto& nd : nodes) { if (nd.left >= 0) sah_n += nd.box.area() / ROOT_AREA; else sah_t += nd.box.area() / ROOT_AREA * nd.count; }
        printf("%-28s: %zu binary nodes, %llu leaf refs (%.3f x), %llu spatial splits | binary: %.2f node boxes + %.2f tri | 8-wide plan: %.2f node steps + %.2f tri tests, cost %.2f\n",
               c.name, nodes.size(), (unsigned long long)n_refs_final, (double)n_refs_final / NT, (unsigned long long)n_spatial, sah_n, sah_t, o.nn[1], o.nt[1], o.c[1]);
        fflush(stdout);
    }
    return 0;
}
