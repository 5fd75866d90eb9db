// libspt_hip.so — C ABI (include/spt_abi.h) over the gfx950 kernels.
// Host code here only validates descriptors, moves the flattened scene into HBM
// and enqueues kernels on one HIP stream; there is no CPU rendering path.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "kernel_list.h"   // the heavy kernel templates: declared here, compiled in inst_*.hip
#include "film_kernels.h"

namespace {

thread_local std::string g_error;

struct AbiError {
    spt_status code;
    std::string msg;
};
[[noreturn]] void fail(spt_status code, const std::string& msg) { throw AbiError{code, msg}; }

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            fail(e_ == hipErrorOutOfMemory ? SPT_ERR_OUT_OF_MEMORY : SPT_ERR_HIP,                    \
                 std::string(#expr) + ": " + hipGetErrorString(e_));                                 \
    } while (0)

int usable_device_count() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

struct DeviceBuffer {   // owning: freed with the scene object (the owner selects the device first)
    void* p = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    void ensure(size_t n) {
        if (n > bytes) alloc(n);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    void upload(const T* src, size_t count) {
        alloc(std::max<size_t>(count * sizeof(T), 16));
        if (count) HIP_CHECK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

uint32_t bvh_depth(const spt_bvh_node* nodes, uint32_t n_nodes, uint32_t root, uint32_t n_items, const char* what) {
    // iterative DFS; also validates indices so the kernels never read out of bounds
    if (n_nodes == 0) return 0;
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.emplace_back(root, 1u);
    uint32_t depth = 0, visited = 0;
    while (!st.empty()) {
        auto [ni, d] = st.back();
        st.pop_back();
        if (ni >= n_nodes) fail(SPT_ERR_INVALID_ARG, std::string(what) + ": node index out of range");
        if (++visited > n_nodes) fail(SPT_ERR_INVALID_ARG, std::string(what) + ": node graph is not a tree");
        depth = std::max(depth, d);
        const spt_bvh_node& nd = nodes[ni];
        if (nd.b & SPT_LEAF_FLAG) {
            uint32_t cnt = nd.b & ~SPT_LEAF_FLAG;
            if ((uint64_t)nd.a + cnt > n_items) fail(SPT_ERR_INVALID_ARG, std::string(what) + ": leaf item range out of bounds");
        } else {
            st.emplace_back(nd.a, d + 1);
            st.emplace_back(nd.b, d + 1);
        }
    }
    return depth;
}

// ---- device-side BLAS ---------------------------------------------------------------------------
// The ABI hands over the caller's trees (the reference side would flatten ITS BvhAccel, whose builder bins
// a primitive by (centroid - its OWN bbox min) / bucket length, src/primitive/bvh.rs:52-57, i.e. by size
// rather than position; libspt_host.so builds a proper binned-SAH tree but always splits down to <= 4).
// A closest / any hit does not depend on the tree, only on the set of triangles, so the library does not
// rely on the caller's tree quality: it builds its own binned-SAH tree per mesh (32 bins, 3 axes, SAH
// leaf termination, leaves of <= 4) over the same triangles.  Node boxes are padded outward (2^-16 of
// the extent + 2^-20 of the magnitude) so that every ray the triangle test accepts also passes the boxes
// above that triangle.  What can differ from a walk of the caller's tree are only the triangle test's own
// false positives for rays grazing a triangle's plane outside its padded box (~1e-7 per ray; none in any
// committed parity case); SPT_REFERENCE_BVH=1 walks the ABI trees instead, and the GPU parity suite is
// green (bit-identical films) in both modes.  Triangles are re-ordered into leaf order in the traversal
// blob; each carries its ABI index in the pad lane of its first vertex (tie-rule key, tri_attr index).
// Measured against libspt_host's trees: cfg2 38.4 -> 39.3 Gsamples/s, cfg5 761 -> 777 Msamples/s.
// Outward padding of a device-side box along one axis.  It has to stay well below Ray::T_MIN_EPS (1e-4):
// a ray leaving a convex object is rejected at the object's root box because it exits the box before
// t_min - with 2^-12 of the extent the cube's rays entered the tree and the fused shade kernel went from
// 2.6 to 3.5 ms (measured).  The slab test only fails for a ray that the triangle test accepts when the
// hit lies within the rounding error of a box EDGE (two slabs barely overlapping), so this small pad
// already makes a wrongly culled hit a ~1e-8-per-ray event.
inline float box_pad(float lo, float hi) {
    return (hi - lo) * 1.52587890625e-5f /* 2^-16 */ + std::max(std::fabs(lo), std::fabs(hi)) * 9.5367431640625e-7f /* 2^-20 */ + 1e-30f;
}
struct SahTri {
    float lo[3], hi[3], c[3];
    uint32_t id;
};
void build_sah(std::vector<SahTri>& t, uint32_t tri_first, uint32_t max_leaf, float traversal_cost, std::vector<spt_bvh_node>& nodes,
               std::vector<uint32_t>& order, const char* what);

void build_sah_blas(const spt_tri_pos* tris, uint32_t tri_first, uint32_t tri_count, std::vector<spt_bvh_node>& nodes,
                    std::vector<uint32_t>& order /* slot (absolute) -> ABI triangle index, filled for this mesh's range */) {
    std::vector<SahTri> t(tri_count);
    for (uint32_t i = 0; i < tri_count; ++i) {
        const spt_tri_pos& p = tris[tri_first + i];
        for (int k = 0; k < 3; ++k) {
            t[i].lo[k] = std::min(p.p0[k], std::min(p.p1[k], p.p2[k]));
            t[i].hi[k] = std::max(p.p0[k], std::max(p.p1[k], p.p2[k]));
            t[i].c[k] = 0.5f * (t[i].lo[k] + t[i].hi[k]);
        }
        t[i].id = tri_first + i;
    }
    // tuning knobs (defaults measured on cfg2 / cfg5; the environment overrides exist for that measurement only)
    uint32_t max_leaf = 4;
    float traversal_cost = 1.2f;   // one wide-node visit (two slab tests) relative to one triangle test
    if (const char* v = std::getenv("SPT_BVH_MAX_LEAF")) max_leaf = (uint32_t)std::min(15, std::max(1, std::atoi(v)));
    if (const char* v = std::getenv("SPT_BVH_TRAVERSAL_COST")) traversal_cost = (float)std::atof(v);
    build_sah(t, tri_first, max_leaf, traversal_cost, nodes, order, "BLAS");
}

// CubicBezier::intersect_ray ACCEPTS a candidate point of the patch that lies within a tolerance of the ray
// (|cross(p - o, d)|^2 < 1e-5 in the patch's object space, bezier.rs:121-131), so a ray that misses the hull of the control
// points by a hair can still "hit" - and the hit's t can lie a hair in front of the box.  Which of those near misses a
// walker sees would then depend on how tight its boxes are and on the order of its visits (fuzz seeds 3017 / 3034 of round
// 2: the streaming walker's quantised boxes against the padded ones, one or two pixels per 600 k samples).  Every box this
// library culls a patch with is therefore widened by the largest distance, in world space, at which the test can
// accept: sqrt(1e-5) over the smallest singular value of cof(M^-1) = sqrt(1e-5) * s1 * s2 (the two largest stretches of
// the instance's M), bounded here by |M|_F^2 / 2.  With that, "tested" is a superset of "can be accepted" for every walker,
// and all of them return what testing every patch returns.  (SPT_REFERENCE_BVH=1 keeps the caller's exact boxes and the
// reference's visit order: that mode reproduces the reference's own loss of such hits.)
float bezier_box_margin(const spt_instance& in) {
    if (in.prim_type != SPT_PRIM_BEZIER) return 0.0f;
    double f2 = 0.0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) f2 += (double)in.fwd[3 * c + r] * (double)in.fwd[3 * c + r];   // the 3 x 3 part (columns), not the translation
    return (float)(0.0031623 * 1.02 * 0.5 * f2 + 1e-6);
}

// The same builder over the instances' world boxes: the device-side TLAS.  An instance visit (transform the ray, walk
// a BLAS) costs far more than a node visit, so leaves hold one instance unless the split is useless.
void build_sah_tlas(const spt_instance* inst, uint32_t n, std::vector<spt_bvh_node>& nodes, std::vector<uint32_t>& order) {
    std::vector<SahTri> t(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float margin = bezier_box_margin(inst[i]);
        for (int k = 0; k < 3; ++k) {
            t[i].lo[k] = inst[i].bmin[k] - margin;
            t[i].hi[k] = inst[i].bmax[k] + margin;
            t[i].c[k] = 0.5f * (t[i].lo[k] + t[i].hi[k]);
        }
        t[i].id = i;
    }
    order.assign(n, 0u);
    build_sah(t, 0u, 2u, 0.125f, nodes, order, "TLAS");
}

// items [0, t.size()) -> a tree appended to `nodes`; leaves index slots tri_first + k, order[slot] = item id
void build_sah(std::vector<SahTri>& t, uint32_t tri_first, uint32_t kMaxLeaf, float kTraversalCost, std::vector<spt_bvh_node>& nodes,
               std::vector<uint32_t>& order, const char* what) {
    const uint32_t tri_count = (uint32_t)t.size();
    constexpr int kBins = 32;
    auto half_area = [](const float* lo, const float* hi) {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Task { uint32_t begin, end, node, depth; };
    std::vector<Task> st;
    const size_t root_index = nodes.size();
    nodes.push_back(spt_bvh_node{});
    st.push_back(Task{0u, tri_count, (uint32_t)nodes.size() - 1u, 0u});
    while (!st.empty()) {
        const Task tk = st.back();
        st.pop_back();
        const uint32_t n = tk.end - tk.begin;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = tk.begin; i < tk.end; ++i)
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], t[i].lo[k]); hi[k] = std::max(hi[k], t[i].hi[k]);
                clo[k] = std::min(clo[k], t[i].c[k]); chi[k] = std::max(chi[k], t[i].c[k]);
            }
        spt_bvh_node nd;
        for (int k = 0; k < 3; ++k) {
            const float pad = box_pad(lo[k], hi[k]);
            nd.bmin[k] = lo[k] - pad;
            nd.bmax[k] = hi[k] + pad;
        }
        auto make_leaf = [&]() {
            nd.a = tri_first + tk.begin;
            nd.b = SPT_LEAF_FLAG | n;
            nodes[tk.node] = nd;
        };
        if (n == 1) { make_leaf(); continue; }
        // best binned split over the three axes
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = 0;
        for (int k = 0; k < 3; ++k) {
            const float ext = chi[k] - clo[k];
            if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
            const float scale = (float)kBins / ext;
            uint32_t cnt[kBins] = {};
            float blo[kBins][3], bhi[kBins][3];
            for (int b = 0; b < kBins; ++b)
                for (int j = 0; j < 3; ++j) { blo[b][j] = INFINITY; bhi[b][j] = -INFINITY; }
            for (uint32_t i = tk.begin; i < tk.end; ++i) {
                int b = std::min(kBins - 1, std::max(0, (int)((t[i].c[k] - clo[k]) * scale)));
                ++cnt[b];
                for (int j = 0; j < 3; ++j) { blo[b][j] = std::min(blo[b][j], t[i].lo[j]); bhi[b][j] = std::max(bhi[b][j], t[i].hi[j]); }
            }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            {
                float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t rc = 0;
                for (int b = kBins - 1; b >= 1; --b) {
                    for (int j = 0; j < 3; ++j) { rl[j] = std::min(rl[j], blo[b][j]); rh[j] = std::max(rh[j], bhi[b][j]); }
                    rc += cnt[b];
                    right_area[b] = rc ? half_area(rl, rh) : 0.0f;
                    right_cnt[b] = rc;
                }
            }
            float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t lc = 0;
            for (int b = 1; b < kBins; ++b) {   // split between bin b-1 and b
                for (int j = 0; j < 3; ++j) { ll[j] = std::min(ll[j], blo[b - 1][j]); lh[j] = std::max(lh[j], bhi[b - 1][j]); }
                lc += cnt[b - 1];
                if (lc == 0 || right_cnt[b] == 0) continue;
                const float cost = half_area(ll, lh) * (float)lc + right_area[b] * (float)right_cnt[b];
                if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = b; }
            }
        }
        const float node_area = half_area(lo, hi);
        uint32_t mid;
        if (best_axis >= 0 && tk.depth < 56u) {
            const float split_cost = kTraversalCost + (node_area > 0.0f ? best_cost / node_area : (float)n);
            if (n <= kMaxLeaf && (float)n <= split_cost) { make_leaf(); continue; }
            const float ext = chi[best_axis] - clo[best_axis];
            const float scale = (float)kBins / ext;
            auto it = std::partition(t.begin() + tk.begin, t.begin() + tk.end, [&](const SahTri& x) {
                int b = std::min(kBins - 1, std::max(0, (int)((x.c[best_axis] - clo[best_axis]) * scale)));
                return b < best_bin;
            });
            mid = (uint32_t)(it - t.begin());
        } else {
            // all centroids coincide (or the tree got too deep): leaf if it fits, else split the range in half
            if (n <= kMaxLeaf) { make_leaf(); continue; }
            mid = tk.begin + n / 2;
            if (best_axis >= 0) {
                const int ax = best_axis;
                std::nth_element(t.begin() + tk.begin, t.begin() + mid, t.begin() + tk.end, [&](const SahTri& x, const SahTri& y) { return x.c[ax] < y.c[ax]; });
            }
        }
        if (mid == tk.begin || mid == tk.end) mid = tk.begin + n / 2;
        nd.a = (uint32_t)nodes.size();
        nd.b = nd.a + 1u;
        nodes[tk.node] = nd;
        nodes.push_back(spt_bvh_node{});
        nodes.push_back(spt_bvh_node{});
        st.push_back(Task{mid, tk.end, nd.b, tk.depth + 1u});
        st.push_back(Task{tk.begin, mid, nd.a, tk.depth + 1u});
    }
    for (uint32_t i = 0; i < tri_count; ++i) order[tri_first + i] = t[i].id;
    if (std::getenv("SPT_DEBUG_BVH")) {
        uint32_t hist[16] = {}, leaves = 0, inner = 0;
        for (size_t k = root_index; k < nodes.size(); ++k) {
            if (nodes[k].b & SPT_LEAF_FLAG) { ++leaves; ++hist[std::min(15u, nodes[k].b & ~SPT_LEAF_FLAG)]; }
            else ++inner;
        }
        std::fprintf(stderr, "[spt] device %s: %u items, %u inner nodes, %u leaves, leaf sizes 1:%u 2:%u 3:%u 4:%u >4:%u\n", what, tri_count, inner, leaves,
                     hist[1], hist[2], hist[3], hist[4], leaves - hist[1] - hist[2] - hist[3] - hist[4]);
    }
}

// Repack one 32-byte-node tree into 64-byte wide nodes (see trace.h).  Returns the index of the
// super-root inside `wide` (in wide-node units).  The first `bfs_nodes` wide nodes are numbered
// breadth-first (they are the ones staged into LDS for large scenes), the subtrees below them
// depth-first so that deep subtrees stay contiguous in memory.
uint32_t build_wide(const spt_bvh_node* nodes, uint32_t root, std::vector<float4>& wide, uint32_t bfs_nodes, const char* what, bool pad_boxes = false) {
    auto leaf_ref = [&](const spt_bvh_node& nd) -> uint32_t {
        uint32_t cnt = nd.b & ~SPT_LEAF_FLAG;
        if (cnt > 15u) fail(SPT_ERR_UNSUPPORTED, std::string(what) + ": BVH leaf with more than 15 items");
        if (nd.a >= (1u << 27)) fail(SPT_ERR_UNSUPPORTED, std::string(what) + ": more than 2^27 items");
        return kLeaf | (cnt << 27) | nd.a;
    };
    auto set_child = [&](uint32_t w, int side, const spt_bvh_node& ch, uint32_t ref) {
        float4* f = &wide[(size_t)w * 4];
        float4 lo = make_float4(ch.bmin[0], ch.bmin[1], ch.bmin[2], 0.0f), hi = make_float4(ch.bmax[0], ch.bmax[1], ch.bmax[2], 0.0f);
        if (pad_boxes && lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z) {
            const float px = box_pad(lo.x, hi.x), py = box_pad(lo.y, hi.y), pz = box_pad(lo.z, hi.z);
            lo.x -= px; lo.y -= py; lo.z -= pz;
            hi.x += px; hi.y += py; hi.z += pz;
        }
        if (side == 0) { f[0].x = lo.x; f[0].y = lo.y; f[0].z = lo.z; f[1].x = hi.x; f[1].y = hi.y; f[1].z = hi.z; std::memcpy(&f[0].w, &ref, 4); }
        else { f[2] = lo; f[3] = hi; std::memcpy(&f[1].w, &ref, 4); }
    };
    auto new_wide = [&]() -> uint32_t {
        uint32_t w = (uint32_t)(wide.size() / 4);
        const float inf = std::numeric_limits<float>::infinity();
        wide.push_back(make_float4(inf, inf, inf, 0.0f));    // empty boxes: never hit
        wide.push_back(make_float4(-inf, -inf, -inf, 0.0f));
        wide.push_back(make_float4(inf, inf, inf, 0.0f));
        wide.push_back(make_float4(-inf, -inf, -inf, 0.0f));
        return w;
    };
    const uint32_t first = (uint32_t)(wide.size() / 4);
    const uint32_t super = new_wide();
    struct Item { uint32_t node, parent, side; };
    // breadth-first part
    std::vector<Item> frontier;
    frontier.push_back(Item{root, super, 0});
    size_t head = 0;
    while (head < frontier.size() && (uint32_t)(wide.size() / 4) - first < bfs_nodes) {
        Item it = frontier[head++];
        const spt_bvh_node& nd = nodes[it.node];
        if (nd.b & SPT_LEAF_FLAG) {
            set_child(it.parent, (int)it.side, nd, leaf_ref(nd));
        } else {
            const uint32_t w = new_wide();
            set_child(it.parent, (int)it.side, nd, w);
            frontier.push_back(Item{nd.a, w, 0});
            frontier.push_back(Item{nd.b, w, 1});
        }
    }
    // depth-first below the frontier
    for (size_t k = head; k < frontier.size(); ++k) {
        std::vector<Item> st;
        st.push_back(frontier[k]);
        while (!st.empty()) {
            Item it = st.back();
            st.pop_back();
            const spt_bvh_node& nd = nodes[it.node];
            if (nd.b & SPT_LEAF_FLAG) {
                set_child(it.parent, (int)it.side, nd, leaf_ref(nd));
            } else {
                const uint32_t w = new_wide();
                set_child(it.parent, (int)it.side, nd, w);
                st.push_back(Item{nd.b, w, 1});
                st.push_back(Item{nd.a, w, 0});
            }
        }
    }
    return super;
}

// Collapse a 2-ary tree of 32-byte nodes into compressed 4-wide nodes (see trace.h).  Returns the ref of
// the root (node index in `out` / 4, or a leaf ref).  Children boxes are quantised OUTWARD and verified
// with exactly the f32 decode arithmetic of node4_test.
// `stack_need` receives the worst-case number of simultaneously pending entries of a near-first walk: a
// node with n children leaves n - 1 of them pending while the first is descended, in whatever order.
uint32_t build_n4(const spt_bvh_node* nodes, uint32_t root, std::vector<float4>& out, uint32_t* stack_need, const char* what) {
    *stack_need = 0;
    auto leaf_ref = [&](const spt_bvh_node& nd) -> uint32_t {
        uint32_t cnt = nd.b & ~SPT_LEAF_FLAG;
        if (cnt > 15u) fail(SPT_ERR_UNSUPPORTED, std::string(what) + ": BVH leaf with more than 15 items");
        if (nd.a >= (1u << 27)) fail(SPT_ERR_UNSUPPORTED, std::string(what) + ": more than 2^27 items");
        return kLeaf | (cnt << 27) | nd.a;
    };
    auto area = [&](const spt_bvh_node& nd) {
        float dx = nd.bmax[0] - nd.bmin[0], dy = nd.bmax[1] - nd.bmin[1], dz = nd.bmax[2] - nd.bmin[2];
        return dx * dy + dy * dz + dz * dx;
    };
    if (nodes[root].b & SPT_LEAF_FLAG) return leaf_ref(nodes[root]);
    struct Item { uint32_t node32, n4; };
    std::vector<Item> st;
    auto alloc = [&]() -> uint32_t {
        uint32_t i = (uint32_t)(out.size() / 4);
        out.resize(out.size() + 4, make_float4(0, 0, 0, 0));
        return i;
    };
    const uint32_t root_n4 = alloc();
    st.push_back(Item{root, root_n4});
    while (!st.empty()) {
        Item it = st.back();
        st.pop_back();
        const spt_bvh_node& par = nodes[it.node32];
        // gather up to 4 children by repeatedly opening the inner child with the largest box
        uint32_t ch[4] = {par.a, par.b, 0, 0};
        uint32_t n = 2;
        while (n < 4) {
            int best = -1;
            float best_area = -1.0f;
            for (uint32_t k = 0; k < n; ++k)
                if (!(nodes[ch[k]].b & SPT_LEAF_FLAG) && area(nodes[ch[k]]) > best_area) { best_area = area(nodes[ch[k]]); best = (int)k; }
            if (best < 0) break;
            const spt_bvh_node& open = nodes[ch[best]];
            ch[best] = open.a;
            ch[n++] = open.b;
        }
        // quantisation frame: p = parent min, scale = 2^e >= extent / 254
        uint32_t eb[3];
        float scale[3];
        for (int k = 0; k < 3; ++k) {
            float ext = par.bmax[k] - par.bmin[k];
            int e = -120;
            if (ext > 0.0f && std::isfinite(ext)) {
                e = (int)std::ceil(std::log2((double)ext / 254.0));
                if (e < -120) e = -120;
                if (e > 120) fail(SPT_ERR_UNSUPPORTED, std::string(what) + ": node extent too large to quantise");
            }
            eb[k] = (uint32_t)(e + 127);
            uint32_t bits = eb[k] << 23;
            std::memcpy(&scale[k], &bits, 4);
        }
        uint32_t qlo[3][4], qhi[3][4], refs[4] = {0, 0, 0, 0};
        for (uint32_t c = 0; c < 4; ++c)
            for (int k = 0; k < 3; ++k) { qlo[k][c] = 255u; qhi[k][c] = 0u; }   // absent child: empty box
        for (uint32_t c = 0; c < n; ++c) {
            const spt_bvh_node& cn = nodes[ch[c]];
            for (int k = 0; k < 3; ++k) {
                const float p = par.bmin[k];
                int lo = (int)std::floor(((double)cn.bmin[k] - (double)p) / (double)scale[k]);
                int hi = (int)std::ceil(((double)cn.bmax[k] - (double)p) / (double)scale[k]);
                lo = std::max(0, std::min(255, lo));
                hi = std::max(0, std::min(255, hi));
                // verify with the device's arithmetic: p + scale * q (f32 multiply, then f32 add)
                auto dec = [&](int q) { float m = scale[k] * (float)q; return p + m; };
                while (lo > 0 && dec(lo) > cn.bmin[k]) --lo;
                while (hi < 255 && dec(hi) < cn.bmax[k]) ++hi;
                if (dec(lo) > cn.bmin[k] || dec(hi) < cn.bmax[k]) fail(SPT_ERR_UNSUPPORTED, std::string(what) + ": cannot quantise a child box conservatively");
                qlo[k][c] = (uint32_t)lo;
                qhi[k][c] = (uint32_t)hi;
            }
            if (cn.b & SPT_LEAF_FLAG) {
                refs[c] = leaf_ref(cn);
            } else {
                refs[c] = alloc();
                st.push_back(Item{ch[c], refs[c]});
            }
        }
        auto pack4 = [](const uint32_t q[4]) { return q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24); };
        auto as_f = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
        float4* f = &out[(size_t)it.n4 * 4];
        f[0] = make_float4(par.bmin[0], par.bmin[1], par.bmin[2], as_f(eb[0] | (eb[1] << 8) | (eb[2] << 16) | (n << 24)));
        f[1] = make_float4(as_f(pack4(qlo[0])), as_f(pack4(qlo[1])), as_f(pack4(qlo[2])), as_f(pack4(qhi[0])));
        f[2] = make_float4(as_f(pack4(qhi[1])), as_f(pack4(qhi[2])), as_f(refs[0]), as_f(refs[1]));
        f[3] = make_float4(as_f(refs[2]), as_f(refs[3]), 0.0f, 0.0f);
    }
    // children are allocated after their parent, so one reverse sweep sees every child before its parent
    const uint32_t end_n4 = (uint32_t)(out.size() / 4);
    std::vector<uint32_t> need(end_n4 - root_n4, 0u);
    for (uint32_t i = end_n4; i-- > root_n4;) {
        const float4* f = &out[(size_t)i * 4];
        uint32_t hdr, refs[4];
        std::memcpy(&hdr, &f[0].w, 4);
        std::memcpy(&refs[0], &f[2].z, 4);
        std::memcpy(&refs[1], &f[2].w, 4);
        std::memcpy(&refs[2], &f[3].x, 4);
        std::memcpy(&refs[3], &f[3].y, 4);
        const uint32_t n = hdr >> 24;
        uint32_t deepest = 0;
        for (uint32_t c = 0; c < n; ++c)
            if (!(refs[c] & kLeaf)) deepest = std::max(deepest, need[refs[c] - root_n4]);
        need[i - root_n4] = n - 1 + deepest;
    }
    *stack_need = need[0];
    return root_n4;
}

}  // namespace

// Scenes with Bezier patches are served by libspt_hip_bez.so: this same source compiled with SPT_WITH_BEZIER=1 (the
// patch test of csrc/hip/bezier.h keeps a 16-frame subdivision stack in scratch memory, and a kernel that can call it
// pays for that scratch on every wave whether or not the scene has patches).  spt_scene_create of the plain library
// opens the other one next to itself and every later call on that scene is passed through.
struct BezierLib {
    void* handle = nullptr;
    spt_status (*create)(const spt_scene_desc*, int32_t, spt_scene**) = nullptr;
    void (*destroy)(spt_scene*) = nullptr;
    spt_status (*render)(const spt_scene*, const spt_camera*, const spt_render_params*, float*, spt_render_stats*) = nullptr;
    spt_status (*render_wait)(const spt_scene*) = nullptr;
    spt_status (*trace_closest)(const spt_scene*, uint32_t, const spt_ray*, spt_hit*) = nullptr;
    spt_status (*trace_any)(const spt_scene*, uint32_t, const spt_ray*, uint8_t*) = nullptr;
    spt_status (*debug_bxdf)(const spt_scene*, int32_t, const spt_material*, uint32_t, uint32_t, const float*, const float*, const uint64_t*, float*, float*,
                             float*, int32_t*) = nullptr;
    const char* (*last_error)(void) = nullptr;
};

struct spt_scene {
    const BezierLib* fwd = nullptr;   // set: `inner` lives in libspt_hip_bez.so and nothing below is used
    spt_scene* inner = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // side stream: k_shadow(b) next to k_extend(b) (see spt_render)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream_copy = nullptr;        // SPT_RENDER_ASYNC: the film's D2H copy, next to the following render's kernels
    hipEvent_t ev_out_ready = nullptr, ev_copy_done = nullptr;
    bool copy_pending = false;                // an asynchronous copy-out of `out` may still be in flight
    bool bez_newton = false;                  // some patch asks for Newton's iteration (the pair kernel only clips)
    // what the last pass with a counter readback saw at bounce 1 (path vertices in all shards); ~0: never seen.  A hint
    // only: it picks between two kernels that compute the same film (k_shade's kLoop)
    uint64_t tail_vertices = ~0ull;
    DScene d{};
    DeviceBuffer tri_pos, tri_attr, instances, meshes, spheres, bezier, surfaces, materials, mediums, lights;
    DeviceBuffer light_props, light_u, light_k, env_px, env_uk, geo;
    bool lds_geo = false;   // traversal geometry small enough to live in LDS (k_*<true>)
    bool swalk = false;     // the streaming walker's tables (stream.h) were built: k_*_stream serve the scene
    size_t lds_bytes = 0;   // dynamic LDS per 256-thread block: traversal stack (+ geometry)
    // render workspace (grown on demand, reused between calls)
    DeviceBuffer qa[5], qb[5], hit_f4, hit_inst, hit_f4_next, hit_inst_next, sh[3], counts, rad, film, first_slot, slot_bits, out;
    DeviceBuffer trace_in, trace_out, visits, inst_class;
    std::mutex mu;
    double bs_center[3] = {0, 0, 0}, bs_radius = 0;  // bounding sphere of all instance boxes
    double world_lo[3] = {0, 0, 0}, world_hi[3] = {0, 0, 0};   // their union
    bool bs_valid = false;
    // per-row screen-space spans (spt_render): the 8 world-space corners of every instance's OBJECT-space box, the spans of
    // the last camera / image size and their device copy
    std::vector<std::array<double, 24>> hull_corners;
    std::vector<int32_t> span_host;          // 2 per image row: first / last pixel that can see an instance (lo > hi: none)
    std::vector<double> span_key;            // camera + image size the spans were made for
    DeviceBuffer row_span;
    // eye-relative copy of the LDS-resident geometry for k_primary<.., kEye> (eye.h): the plain blob and where its parts are
    // (kept on the host), the copy made for the last camera position and the DScene that describes it
    std::vector<float4> host_blob;
    std::vector<std::pair<uint32_t, uint32_t>> blas_range;   // per mesh: its wide nodes [first, end) in units of 4 float4 behind o_blas
    uint32_t wtlas_f4 = 0;
    bool eye_ok = false, eye_valid = false;
    float eye_key[3] = {0, 0, 0};
    DeviceBuffer eye_geo;
    DScene eye_d{};
    size_t eye_lds_bytes = 0;
    bool simple = false;  // Lambert + delta lights only, no emission / environment / media (k_shade<0, .>)
    bool lds_tables = false;  // the shading tables fit LDS behind the geometry (k_shade<.., kTab>)
    bool fused = false;     // k_shade<0, ., kFused> can run: lds_tables and a simple scene
    bool subsurface = false;  // the heavy shade levels are needed: some material has a Subsurface substrate (probe) and / or glints
    bool has_probe = false;   // ... a Subsurface substrate: k_shade<3 | 5, .> (the BSSRDF probe walks the BVH inside the shade kernel)
    bool has_pndf = false;    // ... a position-normal distribution: k_shade<4 | 5, .> (tree walks with private stacks)
    DeviceBuffer ss_cdf;
    DeviceBuffer pndfs, pndf_terms, pndf_nodes, pndf_refs, pndf_roots;   // position-normal distributions (k_shade<3, .> too)
    bool textured = false;  // a material recipe, normal map or emissive map samples textures per hit (k_shade<2, .>)
    DeviceBuffer textures, tex_prog, tex_root, tex_chain, images, image_levels, texels, recipes;
    std::vector<hipEvent_t> events;
    ~spt_scene() {
        if (fwd) { fwd->destroy(inner); return; }
        (void)hipSetDevice(device);
        for (auto e : events) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_out_ready) (void)hipEventDestroy(ev_out_ready);
        if (ev_copy_done) (void)hipEventDestroy(ev_copy_done);
        if (stream_copy) (void)hipStreamDestroy(stream_copy);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

void validate(const spt_scene_desc& s) {
    if (s.abi_version != SPT_ABI_VERSION) fail(SPT_ERR_INVALID_ARG, "scene desc: abi_version mismatch");
    if (s.aggregate > SPT_AGGREGATE_BVH) fail(SPT_ERR_INVALID_ARG, "scene desc: bad aggregate");
    auto need = [](const void* p, uint32_t n, const char* what) {
        if (n && !p) fail(SPT_ERR_INVALID_ARG, std::string("scene desc: null array '") + what + "'");
    };
    need(s.tlas_nodes, s.n_tlas_nodes, "tlas_nodes");
    need(s.instances, s.n_instances, "instances");
    need(s.meshes, s.n_meshes, "meshes");
    need(s.blas_nodes, s.n_blas_nodes, "blas_nodes");
    need(s.tri_pos, s.n_tris, "tri_pos");
    need(s.tri_attr, s.n_tris, "tri_attr");
    need(s.spheres, s.n_spheres, "spheres");
    need(s.bezier_patches, s.n_bezier_patches, "bezier_patches");
    for (uint32_t i = 0; i < s.n_bezier_patches; ++i) {
        const float method = s.bezier_patches[i].cp[0][0][3];
        if (method != 0.0f && method != SPT_BEZIER_NEWTON) fail(SPT_ERR_INVALID_ARG, "scene desc: unknown Bezier intersection method (cp[0][0][3])");
    }
    need(s.surfaces, s.n_surfaces, "surfaces");
    need(s.materials, s.n_materials, "materials");
    need(s.mediums, s.n_mediums, "mediums");
    need(s.lights, s.n_lights, "lights");
    if (s.n_instances && s.aggregate == SPT_AGGREGATE_BVH && s.n_tlas_nodes == 0)
        fail(SPT_ERR_INVALID_ARG, "scene desc: bvh aggregate without TLAS nodes");
    for (uint32_t i = 0; i < s.n_instances; ++i) {
        const spt_instance& in = s.instances[i];
        if (in.prim_type == SPT_PRIM_SPHERE) {
            if (in.prim_id >= s.n_spheres) fail(SPT_ERR_INVALID_ARG, "scene desc: instance sphere index out of range");
        } else if (in.prim_type == SPT_PRIM_MESH) {
            if (in.prim_id >= s.n_meshes) fail(SPT_ERR_INVALID_ARG, "scene desc: instance mesh index out of range");
        } else if (in.prim_type == SPT_PRIM_BEZIER) {
            if (!SPT_WITH_BEZIER) fail(SPT_ERR_UNSUPPORTED, "scene desc: Bezier instances are served by libspt_hip_bez.so");   // not reached: spt_scene_create forwards
            if (in.prim_id >= s.n_bezier_patches) fail(SPT_ERR_INVALID_ARG, "scene desc: instance Bezier patch index out of range");
            // CubicBezier::sample / pdf / surface_area are `unimplemented!` in the reference (bezier.rs:180-190)
            if (in.light >= 0) fail(SPT_ERR_UNSUPPORTED, "scene desc: a Bezier patch cannot be a shape light");
        } else {
            fail(SPT_ERR_INVALID_ARG, "scene desc: bad instance prim_type");
        }
        if (in.surface >= s.n_surfaces) fail(SPT_ERR_INVALID_ARG, "scene desc: instance surface index out of range");
        if (in.light >= (int32_t)s.n_lights) fail(SPT_ERR_INVALID_ARG, "scene desc: instance light index out of range");
        if (in.light >= 0 && (s.lights[in.light].type != SPT_LIGHT_SHAPE || s.lights[in.light].instance != i))
            fail(SPT_ERR_INVALID_ARG, "scene desc: instance light does not name the shape light of this instance");
        // pdf_shape_light of the power_is sampler looks the instance up in its light map (power_is.rs:84-86; the reference
        // panics on a missing key): an emissive instance that is not a light would index light_alias.props[-1]
        if (s.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS && in.light < 0) {
            const spt_surface& sf = s.surfaces[in.surface];
            if (0.299f * sf.emissive[0] + 0.587f * sf.emissive[1] + 0.114f * sf.emissive[2] > 0.0f)
                fail(SPT_ERR_INVALID_ARG, "scene desc: power_is light sampler and an emissive instance without a shape light");
        }
    }
    for (uint32_t i = 0; i < s.n_meshes; ++i) {
        const spt_mesh& m = s.meshes[i];
        if (m.root >= s.n_blas_nodes || (uint64_t)m.tri_first + m.tri_count > s.n_tris || m.tri_count == 0)
            fail(SPT_ERR_INVALID_ARG, "scene desc: mesh ranges out of bounds");
    }
    for (uint32_t i = 0; i < s.n_surfaces; ++i) {
        if (s.surfaces[i].material >= s.n_materials) fail(SPT_ERR_INVALID_ARG, "scene desc: surface material out of range");
        if (s.surfaces[i].inside_medium >= (int32_t)s.n_mediums) fail(SPT_ERR_INVALID_ARG, "scene desc: surface medium out of range");
        if (s.surfaces[i].inside_medium >= 254) fail(SPT_ERR_UNSUPPORTED, "scene desc: more than 254 mediums");
    }
    for (uint32_t i = 0; i < s.n_materials; ++i) {
        if (s.materials[i].bxdf > SPT_BXDF_SPECULAR_PLASTIC) fail(SPT_ERR_INVALID_ARG, "scene desc: unknown bxdf tag");   // SPT_BXDF_PNDF_CONDUCTOR only exists per hit
        if (s.materials[i].recipe > s.n_material_recipes) fail(SPT_ERR_INVALID_ARG, "scene desc: material recipe out of range");
    }
    need(s.textures, s.n_textures, "textures");
    need(s.images, s.n_images, "images");
    need(s.image_levels, s.n_image_levels, "image_levels");
    need(s.texels, s.n_texels, "texels");
    need(s.material_recipes, s.n_material_recipes, "material_recipes");
    for (uint32_t i = 0; i < s.n_images; ++i) {
        const spt_image& im = s.images[i];
        if (im.n_levels == 0 || (uint64_t)im.first_level + im.n_levels > s.n_image_levels) fail(SPT_ERR_INVALID_ARG, "scene desc: image level range out of bounds");
    }
    for (uint32_t i = 0; i < s.n_image_levels; ++i) {
        const spt_image_level& L = s.image_levels[i];
        if (L.width == 0 || L.height == 0 || (uint64_t)L.width * L.height > 0x7fffffffull ||
            (uint64_t)L.first_texel + (uint64_t)L.width * L.height > s.n_texels)
            fail(SPT_ERR_INVALID_ARG, "scene desc: image level texels out of bounds");
    }
    for (uint32_t i = 0; i < s.n_textures; ++i) {
        const spt_texture& t = s.textures[i];
        if (t.type > SPT_TEX_MODIFIER) fail(SPT_ERR_INVALID_ARG, "scene desc: unknown texture type");
        if (t.type == SPT_TEX_IMAGE && t.image >= s.n_images) fail(SPT_ERR_INVALID_ARG, "scene desc: texture image out of range");
        const bool unary = t.type == SPT_TEX_SRGB || t.type == SPT_TEX_MODIFIER, binary = t.type >= SPT_TEX_ADD && t.type <= SPT_TEX_DIV;
        if ((unary || binary) && t.a >= i) fail(SPT_ERR_INVALID_ARG, "scene desc: texture child must precede its parent");
        if (binary && t.b >= i) fail(SPT_ERR_INVALID_ARG, "scene desc: texture child must precede its parent");
        if (t.type == SPT_TEX_MODIFIER && (t.mode > SPT_TEXMODE_BITANGENT || t.wrap > SPT_TEXWRAP_MIRROR_CLAMP))
            fail(SPT_ERR_INVALID_ARG, "scene desc: bad texture input mode / wrap");
    }
    for (uint32_t i = 0; i < s.n_material_recipes; ++i) {
        const spt_material_recipe& r = s.material_recipes[i];
        if (r.type > SPT_MAT_PNDF_PLASTIC || r.rough_chan > SPT_CHAN_A || r.metal_chan > SPT_CHAN_A) fail(SPT_ERR_INVALID_ARG, "scene desc: bad material recipe");
        for (int k = 0; k < 4; ++k) {
            if (r.type >= SPT_MAT_PNDF_CONDUCTOR && k == 1) {
                if (r.tex[k] >= s.n_pndfs) fail(SPT_ERR_INVALID_ARG, "scene desc: material recipe P-NDF out of range");
            } else if (r.tex[k] >= s.n_textures) {
                fail(SPT_ERR_INVALID_ARG, "scene desc: material recipe texture out of range");
            }
        }
    }
    // position-normal distributions: every index the per-hit tree walks of include/spt_pndf.h follow
    need(s.pndfs, s.n_pndfs, "pndfs");
    need(s.pndf_terms, s.n_pndf_terms, "pndf_terms");
    need(s.pndf_nodes, s.n_pndf_nodes, "pndf_nodes");
    need(s.pndf_refs, s.n_pndf_refs, "pndf_refs");
    need(s.pndf_roots, s.n_pndf_roots, "pndf_roots");
    for (uint32_t i = 0; i < s.n_pndf_refs; ++i)
        if (s.pndf_refs[i] >= s.n_pndf_terms) fail(SPT_ERR_INVALID_ARG, "scene desc: P-NDF term reference out of range");
    for (uint32_t i = 0; i < s.n_pndf_nodes; ++i) {
        const spt_pndf_node& n = s.pndf_nodes[i];
        const bool leaf = n.lc == 0xffffffffu;
        // children behind their parent: no walk can cycle
        if (!leaf && (n.lc <= i || n.rc <= i || n.lc >= s.n_pndf_nodes || n.rc >= s.n_pndf_nodes)) fail(SPT_ERR_INVALID_ARG, "scene desc: P-NDF node children out of order");
        if (n.start > n.end || n.end > s.n_pndf_refs) fail(SPT_ERR_INVALID_ARG, "scene desc: P-NDF node range out of bounds");
    }
    {
        // the leaves' ranges are relative to their tree's first ref: one pass over every tree
        // ... and every node belongs to exactly ONE tree and is reached once (a shared child would make this walk - and a
        // DAG-shaped descriptor the device's - exponential), at a depth the walks' fixed stacks hold: spt_pndf_calc /
        // spt_pndf_uv_walk pop one node and push two, so an inner node at depth d (root = 1) leaves d + 1 entries pending;
        // beyond SPT_PNDF_STACK they would drop subtrees silently (a wrong density, not an error)
        std::vector<std::pair<uint32_t, uint32_t>> todo;
        std::vector<uint8_t> seen(s.n_pndf_nodes, 0);
        auto check_tree = [&](uint32_t root, uint32_t first_ref) {
            if (root == 0xffffffffu) return;
            if (root >= s.n_pndf_nodes || first_ref > s.n_pndf_refs) fail(SPT_ERR_INVALID_ARG, "scene desc: P-NDF tree root out of range");
            todo.assign(1, std::make_pair(root, 1u));
            while (!todo.empty()) {
                const uint32_t ni = todo.back().first, depth = todo.back().second;
                todo.pop_back();
                if (seen[ni]) fail(SPT_ERR_INVALID_ARG, "scene desc: a P-NDF node is reachable twice (trees must not share nodes)");
                seen[ni] = 1;
                const spt_pndf_node& n = s.pndf_nodes[ni];
                if ((uint64_t)first_ref + n.end > s.n_pndf_refs) fail(SPT_ERR_INVALID_ARG, "scene desc: P-NDF leaf range out of bounds");
                if (n.lc != 0xffffffffu) {
                    if (depth + 1u > SPT_PNDF_STACK) fail(SPT_ERR_UNSUPPORTED, "scene desc: P-NDF tree deeper than the walks' stack (" + std::to_string(SPT_PNDF_STACK) + " pending entries)");
                    todo.emplace_back(n.lc, depth + 1u);
                    todo.emplace_back(n.rc, depth + 1u);
                }
            }
        };
        for (uint32_t i = 0; i < s.n_pndfs; ++i) {
            const spt_pndf& pd = s.pndfs[i];
            if (pd.n_terms == 0 || (uint64_t)pd.first_term + pd.n_terms > s.n_pndf_terms) fail(SPT_ERR_INVALID_ARG, "scene desc: P-NDF term range out of bounds");
            if (pd.s_block_count == 0 || pd.s_block_count > 4096u || (uint64_t)pd.first_root + 2ull * pd.s_block_count * pd.s_block_count > s.n_pndf_roots)
                fail(SPT_ERR_INVALID_ARG, "scene desc: P-NDF block table out of bounds");
            for (uint32_t b = 0; b < pd.s_block_count * pd.s_block_count; ++b) check_tree(s.pndf_roots[pd.first_root + 2u * b], s.pndf_roots[pd.first_root + 2u * b + 1u]);
            check_tree(pd.uv_root, pd.uv_first_ref);
        }
    }
    for (uint32_t i = 0; i < s.n_surfaces; ++i)
        if (s.surfaces[i].normal_map > s.n_textures || s.surfaces[i].emissive_map > s.n_textures)
            fail(SPT_ERR_INVALID_ARG, "scene desc: surface map out of range");
    for (uint32_t i = 0; i < s.n_lights; ++i) {
        const spt_light& l = s.lights[i];
        if (l.type > SPT_LIGHT_ENV) fail(SPT_ERR_INVALID_ARG, "scene desc: unknown light type");
        if (l.type == SPT_LIGHT_SHAPE && l.instance >= s.n_instances) fail(SPT_ERR_INVALID_ARG, "scene desc: shape light instance out of range");
        if (l.type == SPT_LIGHT_ENV && (s.env.width == 0 || s.env.height == 0)) fail(SPT_ERR_INVALID_ARG, "scene desc: env light without env map");
    }
    if (s.env_light_index >= (int32_t)s.n_lights) fail(SPT_ERR_INVALID_ARG, "scene desc: env_light_index out of range");
    if (s.light_sampler > SPT_LIGHT_SAMPLER_POWER_IS) fail(SPT_ERR_INVALID_ARG, "scene desc: bad light_sampler");
    if (s.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS && s.n_lights) {
        if (s.light_alias.n != s.n_lights || !s.light_alias.props || !s.light_alias.u || !s.light_alias.k)
            fail(SPT_ERR_INVALID_ARG, "scene desc: power_is sampler needs an alias table over the lights");
        for (uint32_t i = 0; i < s.n_lights; ++i)
            if (s.light_alias.k[i] >= s.n_lights) fail(SPT_ERR_INVALID_ARG, "scene desc: light alias index out of range");
    }
    if (s.env.width || s.env.height) {
        uint64_t n = (uint64_t)s.env.width * s.env.height;
        if (n == 0 || n > 0x7fffffffull) fail(SPT_ERR_INVALID_ARG, "scene desc: bad env size");
        if (!s.env.texels || !s.env.alias.props || !s.env.alias.u || !s.env.alias.k || s.env.alias.n != n)
            fail(SPT_ERR_INVALID_ARG, "scene desc: env map needs texels and an alias table");
        for (uint64_t i = 0; i < n; ++i)
            if (s.env.alias.k[i] >= n) fail(SPT_ERR_INVALID_ARG, "scene desc: env alias index out of range");
    }
}

uint32_t shard_row_count(const spt_render_params& p) {
    uint32_t sc = p.shard_count ? p.shard_count : 1u, sr = p.strip_rows ? p.strip_rows : 1u;
    uint32_t rows = 0;
    for (uint32_t j = 0; j < p.height; ++j)
        if ((j / sr) % sc == p.shard_index) ++rows;
    return rows;
}

constexpr uint32_t kBlock = 256;
// Grid of the queue kernels (grid-stride over a queue shard).  Sizing it to exactly the resident workgroups of
// each kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor: 768 for the 144-VGPR fused shade kernel) was MEASURED
// no better (cfg2 60.5 / 61.8 / 62.1 Gsamples/s for 1 / 2 / 3 resident rounds vs 62.4 with this fixed grid, cfg4 4.49
// vs 4.61): items cost very different amounts, so more, smaller work shares balance better than an exact fit.
constexpr uint32_t kPersistentBlocks = 2048;
// bounce-1 vertices per pass below which the fused pipeline stops launching per bounce (a lane that loops over its path
// wastes the lanes whose paths ended; with this few vertices that costs microseconds, the launches it saves ~0.1 ms)
constexpr uint64_t kTailLoopBelow = 4ull << 20;

}  // namespace

extern "C" uint32_t spt_abi_version(void);

#if !SPT_WITH_BEZIER
namespace {
const BezierLib* bezier_lib() {
    static BezierLib lib;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        Dl_info info;
        if (!dladdr(reinterpret_cast<void*>(&spt_abi_version), &info) || !info.dli_fname) { err = "dladdr failed"; return; }
        std::string path(info.dli_fname);
        const size_t slash = path.find_last_of('/');
        path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "libspt_hip_bez.so";
        void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { const char* e = dlerror(); err = e ? e : "dlopen failed"; return; }
        lib.create = reinterpret_cast<decltype(lib.create)>(dlsym(h, "spt_scene_create"));
        lib.destroy = reinterpret_cast<decltype(lib.destroy)>(dlsym(h, "spt_scene_destroy"));
        lib.render = reinterpret_cast<decltype(lib.render)>(dlsym(h, "spt_render"));
        lib.render_wait = reinterpret_cast<decltype(lib.render_wait)>(dlsym(h, "spt_render_wait"));
        lib.trace_closest = reinterpret_cast<decltype(lib.trace_closest)>(dlsym(h, "spt_trace_closest"));
        lib.trace_any = reinterpret_cast<decltype(lib.trace_any)>(dlsym(h, "spt_trace_any"));
        lib.debug_bxdf = reinterpret_cast<decltype(lib.debug_bxdf)>(dlsym(h, "spt_debug_bxdf"));
        lib.last_error = reinterpret_cast<decltype(lib.last_error)>(dlsym(h, "spt_last_error"));
        auto version = reinterpret_cast<uint32_t (*)(void)>(dlsym(h, "spt_abi_version"));
        if (!lib.create || !lib.destroy || !lib.render || !lib.render_wait || !lib.trace_closest || !lib.trace_any || !lib.debug_bxdf || !lib.last_error || !version || version() != SPT_ABI_VERSION) {
            err = path + " does not export ABI version " + std::to_string(SPT_ABI_VERSION);
            return;
        }
        lib.handle = h;
    });
    if (!lib.handle) fail(SPT_ERR_UNSUPPORTED, "the scene has Bezier patches and libspt_hip_bez.so could not be loaded: " + err);
    return &lib;
}
}  // namespace
#endif

extern "C" {

const char* spt_last_error(void) { return g_error.c_str(); }
uint32_t spt_abi_version(void) { return SPT_ABI_VERSION; }

spt_status spt_device_count(int32_t* count) {
    if (!count) { g_error = "device_count: null argument"; return SPT_ERR_INVALID_ARG; }
    *count = usable_device_count();
    return SPT_OK;
}

spt_status spt_shard_rows(const spt_render_params* params, uint32_t* rows) {
    if (!params || !rows) { g_error = "shard_rows: null argument"; return SPT_ERR_INVALID_ARG; }
    *rows = shard_row_count(*params);
    return SPT_OK;
}

spt_status spt_scene_create(const spt_scene_desc* desc, int32_t device, spt_scene** out) {
    if (!desc || !out) { g_error = "scene_create: null argument"; return SPT_ERR_INVALID_ARG; }
    *out = nullptr;
    spt_scene* sc = nullptr;
    try {
#if !SPT_WITH_BEZIER
        bool has_patch_instance = false;   // a patch that no instance uses (a primitives library) does not count
        if (desc->n_bezier_patches > 0 && desc->instances)
            for (uint32_t i = 0; i < desc->n_instances && !has_patch_instance; ++i) has_patch_instance = desc->instances[i].prim_type == SPT_PRIM_BEZIER;
        if (has_patch_instance) {   // see BezierLib
            const BezierLib* lib = bezier_lib();
            spt_scene* inner = nullptr;
            const spt_status st = lib->create(desc, device, &inner);
            if (st != SPT_OK) fail(st, lib->last_error());
            sc = new spt_scene();
            sc->fwd = lib;
            sc->inner = inner;
            *out = sc;
            return SPT_OK;
        }
#endif
        validate(*desc);
        int n = usable_device_count();
        if (n <= 0) fail(SPT_ERR_NO_DEVICE, "no HIP device is visible: libspt_hip has no CPU fallback");
        if (device < 0 || device >= n) fail(SPT_ERR_NO_DEVICE, "device index out of range");
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            fail(SPT_ERR_NO_DEVICE, std::string("device is '") + prop.gcnArchName + "', the kernels are built for gfx950 only");
        HIP_CHECK(hipSetDevice(device));
        sc = new spt_scene();
        sc->device = device;
        HIP_CHECK(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&sc->stream2, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&sc->stream_copy, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_out_ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_copy_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_join, hipEventDisableTiming));
        const spt_scene_desc& s = *desc;
        // stack need: reference-order traversal holds at most depth+1 entries per level of nesting
        uint32_t tlas_depth = 0, blas_depth = 0;
        if (s.aggregate == SPT_AGGREGATE_BVH) tlas_depth = bvh_depth(s.tlas_nodes, s.n_tlas_nodes, 0, s.n_instances, "tlas");
        for (uint32_t i = 0; i < s.n_meshes; ++i) {
            // each BLAS must index triangles inside its own mesh range
            blas_depth = std::max(blas_depth, bvh_depth(s.blas_nodes, s.n_blas_nodes, s.meshes[i].root, s.n_tris, "blas"));
        }
        // near-first traversal pushes at most one (far) child per 2-wide level (4-wide trees: see build_n4)
        const bool own_bvh = std::getenv("SPT_REFERENCE_BVH") == nullptr;   // see build_sah_blas
        uint32_t cap = tlas_depth + blas_depth + 2;
        if (!own_bvh && cap > kLdsStack + kSpillStack) fail(SPT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack (48 levels)");
        if (tlas_depth + 2 > kLdsStack + kSpillStack) fail(SPT_ERR_UNSUPPORTED, "TLAS deeper than the traversal stack (48 levels)");
        sc->tri_pos.upload(s.tri_pos, s.n_tris);
        sc->tri_attr.upload(s.tri_attr, s.n_tris);
        sc->instances.upload(s.instances, s.n_instances);
        sc->meshes.upload(s.meshes, s.n_meshes);
        sc->spheres.upload(s.spheres, s.n_spheres);
        sc->bezier.upload(s.bezier_patches, s.n_bezier_patches);
        for (uint32_t i = 0; i < s.n_bezier_patches; ++i) sc->bez_newton = sc->bez_newton || s.bezier_patches[i].cp[0][0][3] != 0.0f;
        sc->surfaces.upload(s.surfaces, s.n_surfaces);
        sc->materials.upload(s.materials, s.n_materials);
        sc->mediums.upload(s.mediums, s.n_mediums);
        sc->lights.upload(s.lights, s.n_lights);
        const bool pis = s.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS;
        sc->light_props.upload(s.light_alias.props, pis ? s.n_lights : 0);
        sc->light_u.upload(s.light_alias.u, pis ? s.n_lights : 0);
        sc->light_k.upload(s.light_alias.k, pis ? s.n_lights : 0);
        size_t ne = (size_t)s.env.width * s.env.height;
        {
            std::vector<float4> px(ne);
            std::vector<uint2> uk(ne);
            for (size_t i = 0; i < ne; ++i) {
                px[i] = make_float4(s.env.texels[3 * i], s.env.texels[3 * i + 1], s.env.texels[3 * i + 2], s.env.alias.props[i]);
                uint32_t ub;
                std::memcpy(&ub, &s.env.alias.u[i], 4);
                uk[i] = make_uint2(ub, s.env.alias.k[i]);
            }
            sc->env_px.upload(px.data(), ne);
            sc->env_uk.upload(uk.data(), ne);
        }
        DScene& d = sc->d;
        d.tri_pos = sc->tri_pos.as<float4>();
        d.tri_attr = sc->tri_attr.as<float4>();
        d.instances = sc->instances.as<float4>();
        d.meshes = sc->meshes.as<uint4>();
        d.spheres = sc->spheres.as<float4>();
        d.bez = sc->bezier.as<float4>();
        d.surfaces = sc->surfaces.as<spt_surface>();
        d.materials = sc->materials.as<spt_material>();
        d.mediums = sc->mediums.as<spt_medium>();
        d.lights = sc->lights.as<spt_light>();
        d.light_props = sc->light_props.as<float>();
        d.light_u = sc->light_u.as<float>();
        d.light_k = sc->light_k.as<uint32_t>();
        d.env_px = sc->env_px.as<float4>();
        d.env_uk = sc->env_uk.as<uint2>();
        d.n_tlas_nodes = s.n_tlas_nodes;
        d.n_instances = s.n_instances;
        d.n_lights = s.n_lights;
        d.n_meshes = s.n_meshes;
        d.aggregate = s.aggregate;
        d.light_sampler = s.light_sampler;
        d.env_light_index = s.env_light_index;
        d.env_w = s.env.width;
        d.env_h = s.env.height;
        for (int i = 0; i < 3; ++i) d.env_scale[i] = s.env.scale[i];
        d.stack_cap = cap;
        d.fast_slab = own_bvh ? 1u : 0u;
        // the class a hit on each instance is queued under for the shade stage of bounce >= 1 (kernels.h, kClasses): by the code
        // path its material takes through mat_sample / mat_eval / mat_pdf and by whether it has a light sample at all
        std::vector<uint8_t> cls(std::max<uint32_t>(s.n_instances, 1u), 0);
        for (uint32_t i = 0; i < s.n_instances; ++i) {
            const spt_material& m = s.materials[s.surfaces[s.instances[i].surface].material];
            uint32_t b = m.bxdf;
            if (m.recipe != 0u) {      // evaluated per hit: the kind the recipe usually resolves to
                switch (s.material_recipes[m.recipe - 1u].type) {
                case SPT_MAT_LAMBERT: b = SPT_BXDF_LAMBERT; break;
                case SPT_MAT_CONDUCTOR: b = SPT_BXDF_MICROFACET_CONDUCTOR; break;
                case SPT_MAT_DIELECTRIC: b = SPT_BXDF_MICROFACET_DIELECTRIC; break;
                default: b = SPT_BXDF_MICROFACET_PLASTIC; break;
                }
            }
            switch (b) {
            case SPT_BXDF_LAMBERT: cls[i] = 0; break;
            case SPT_BXDF_MICROFACET_CONDUCTOR: cls[i] = 1; break;
            case SPT_BXDF_SPECULAR_CONDUCTOR: cls[i] = 2; break;
            case SPT_BXDF_MICROFACET_DIELECTRIC: cls[i] = 3; break;
            case SPT_BXDF_SPECULAR_DIELECTRIC: cls[i] = 4; break;
            case SPT_BXDF_PSEUDO: cls[i] = 5; break;
            default: cls[i] = 6; break;       // the plastic / PBR lobes, glints
            }
        }
        sc->inst_class.upload(cls.data(), cls.size());
        d.inst_class = sc->inst_class.as<uint8_t>();
        {
            // one float4 blob for everything the traversal touches: [tlas | instances | meshes | spheres | blas | tri]
            std::vector<float4> blob;
            auto append = [&](const void* src, size_t bytes) {
                uint32_t off = (uint32_t)blob.size();
                size_t n4 = (bytes + 15) / 16;
                blob.resize(blob.size() + n4, make_float4(0, 0, 0, 0));
                if (bytes) std::memcpy(&blob[off], src, bytes);
                return off;
            };
            std::vector<float4> wtlas;
            d.tlas_root = 0;
            for (int k = 0; k < 3; ++k) { d.tlas_lo[k] = 0.0f; d.tlas_hi[k] = 0.0f; }
            // TLAS leaf slot -> instance index: identity for the caller's tree (its leaves index the instance array) and
            // for a GROUP aggregate, the leaf order of the device-built tree otherwise
            std::vector<uint32_t> tlas_order(s.n_instances);
            for (uint32_t i = 0; i < s.n_instances; ++i) tlas_order[i] = i;
            std::vector<spt_bvh_node> own_tlas;
            if (s.aggregate == SPT_AGGREGATE_BVH && s.n_tlas_nodes) {
                if (own_bvh && s.n_instances) {
                    build_sah_tlas(s.instances, s.n_instances, own_tlas, tlas_order);   // boxes padded by the builder
                    tlas_depth = bvh_depth(own_tlas.data(), (uint32_t)own_tlas.size(), 0, s.n_instances, "device tlas");
                    if (tlas_depth + 2 > kLdsStack + kSpillStack) fail(SPT_ERR_UNSUPPORTED, "device TLAS deeper than the traversal stack (48 levels)");
                }
                const spt_bvh_node* tlas_src = own_tlas.empty() ? s.tlas_nodes : own_tlas.data();
                const uint32_t sup = build_wide(tlas_src, 0, wtlas, 0xffffffffu, "tlas", false);
                std::memcpy(&d.tlas_root, &wtlas[(size_t)sup * 4].w, 4);      // left child of the super-root = real root
                for (int k = 0; k < 3; ++k) { d.tlas_lo[k] = tlas_src[0].bmin[k]; d.tlas_hi[k] = tlas_src[0].bmax[k]; }
            }
            // BLAS: 2-wide full-precision nodes when the whole scene fits LDS, compressed 4-wide nodes otherwise
            // own binned-SAH BLAS per mesh (see build_sah_blas), or the ABI trees as they are
            std::vector<spt_bvh_node> own_nodes;
            std::vector<uint32_t> own_roots(s.n_meshes, 0u), tri_order(s.n_tris);
            for (uint32_t i = 0; i < s.n_tris; ++i) tri_order[i] = i;
            if (own_bvh) {
                for (uint32_t i = 0; i < s.n_meshes; ++i) {
                    own_roots[i] = (uint32_t)own_nodes.size();
                    build_sah_blas(s.tri_pos, s.meshes[i].tri_first, s.meshes[i].tri_count, own_nodes, tri_order);
                    const uint32_t depth = bvh_depth(own_nodes.data(), (uint32_t)own_nodes.size(), own_roots[i], s.n_tris, "device blas");
                    if (tlas_depth + depth + 2 > kLdsStack + kSpillStack) fail(SPT_ERR_UNSUPPORTED, "device BVH deeper than the traversal stack (48 levels)");
                }
            }
            const spt_bvh_node* blas_src = own_bvh ? own_nodes.data() : s.blas_nodes;
            // triangles in leaf order of the tree that is walked, ABI index in the first vertex's pad lane
            std::vector<spt_tri_pos> tri_blob(s.n_tris);
            for (uint32_t i = 0; i < s.n_tris; ++i) {
                const spt_tri_pos& src = s.tri_pos[tri_order[i]];
                spt_tri_pos& dst = tri_blob[i];
                dst = src;
                for (int k = 0; k < 3; ++k) { dst.p1[k] = src.p1[k] - src.p0[k]; dst.p2[k] = src.p2[k] - src.p0[k]; }   // e1, e2 of triangle.rs:125-126
                std::memcpy(&dst.pad0, &tri_order[i], 4);
            }
            auto assemble = [&](bool n4) {
                blob.clear();
                sc->blas_range.clear();
                std::vector<float4> wblas;
                std::vector<float4> mesh_rec((size_t)s.n_meshes * 2, make_float4(0, 0, 0, 0));   // (root.lo, root ref) (root.hi, -)
                uint32_t max_blas_need = 0;
                for (uint32_t i = 0; i < s.n_meshes; ++i) {
                    const uint32_t mesh_root = own_bvh ? own_roots[i] : s.meshes[i].root;
                    const spt_bvh_node& rn = blas_src[mesh_root];
                    uint32_t root_ref;
                    if (n4) {
                        uint32_t need = 0;
                        root_ref = build_n4(blas_src, mesh_root, wblas, &need, "blas");
                        max_blas_need = std::max(max_blas_need, need);
                        if (tlas_depth + need + 2 > kLdsStack + kSpillStack)
                            fail(SPT_ERR_UNSUPPORTED, "4-wide BVH needs more than the traversal stack (48 entries)");
                    } else {
                        const uint32_t first_node = (uint32_t)(wblas.size() / 4);
                        const uint32_t sup = build_wide(blas_src, mesh_root, wblas, 0u, "blas");
                        std::memcpy(&root_ref, &wblas[(size_t)sup * 4].w, 4);
                        sc->blas_range.emplace_back(first_node, (uint32_t)(wblas.size() / 4));
                    }
                    float rf;
                    std::memcpy(&rf, &root_ref, 4);
                    mesh_rec[2 * i] = make_float4(rn.bmin[0], rn.bmin[1], rn.bmin[2], rf);
                    // (flat.h: the mesh's range in the blob's triangle copy, when it fits 16 + 16 bits)
                    const spt_mesh& mm = s.meshes[i];
                    const uint32_t range = (mm.tri_first < 65536u && mm.tri_count < 65536u) ? (mm.tri_first | (mm.tri_count << 16)) : 0u;
                    float range_f;
                    std::memcpy(&range_f, &range, 4);
                    mesh_rec[2 * i + 1] = make_float4(rn.bmax[0], rn.bmax[1], rn.bmax[2], range_f);
                }
                // streaming walker (stream.h): a 4-wide TLAS behind the BLAS nodes (refs of both levels index one array) and
                // 96-byte instance entry records in its leaf order.  Its boxes only cull (quantised outward, relaxed
                // test), so one tree serves both BVH modes and a GROUP aggregate alike.
                sc->swalk = false;
                d.s_root = 0u;
                std::vector<float4> sinst;
                if (n4 && s.n_instances) {
                    const size_t blas_f4 = wblas.size();
                    try {
                        std::vector<spt_bvh_node> group_tlas;
                        std::vector<uint32_t> s_order = tlas_order;
                        const spt_bvh_node* src = nullptr;
                        if (s.aggregate == SPT_AGGREGATE_BVH && s.n_tlas_nodes) {
                            src = own_tlas.empty() ? s.tlas_nodes : own_tlas.data();
                        } else {
                            build_sah_tlas(s.instances, s.n_instances, group_tlas, s_order);
                            src = group_tlas.data();
                            for (int k = 0; k < 3; ++k) { d.tlas_lo[k] = src[0].bmin[k]; d.tlas_hi[k] = src[0].bmax[k]; }
                        }
                        uint32_t need_t = 0;
                        d.s_root = build_n4(src, 0, wblas, &need_t, "tlas");
                        // + 1: a TLAS leaf of several instances keeps its remaining instances as one extra entry
                        if (need_t + max_blas_need + 3 > kLdsStack + kSpillStack) fail(SPT_ERR_UNSUPPORTED, "4-wide TLAS + BLAS need more than the traversal stack");
                        sinst.assign((size_t)s.n_instances * 6, make_float4(0, 0, 0, 0));
                        auto as_f = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
                        for (uint32_t slot = 0; slot < s.n_instances; ++slot) {
                            const uint32_t ii = s_order[slot];
                            const spt_instance& in = s.instances[ii];
                            float4* r = &sinst[(size_t)slot * 6];
                            std::memcpy(r, in.inv, 48);
                            r[3] = make_float4(as_f(ii), as_f(in.prim_type), as_f(in.prim_id), 0.0f);
                            if (in.prim_type == SPT_PRIM_MESH) { r[4] = mesh_rec[2 * in.prim_id]; r[5] = mesh_rec[2 * in.prim_id + 1]; }
                            else if (in.prim_type == SPT_PRIM_SPHERE) { const spt_sphere& sp = s.spheres[in.prim_id]; r[4] = make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius); }
                        }
                        sc->swalk = std::getenv("SPT_NO_STREAM") == nullptr;
                        // (patches under the caller's exact boxes: only the reference's visit order reproduces which near misses it loses)
                        if (!own_bvh && s.n_bezier_patches != 0) sc->swalk = false;
                    } catch (const AbiError&) {
                        wblas.resize(blas_f4);   // e.g. an instance box too large to quantise: the walkers of trace.h serve the scene
                        sinst.clear();
                        sc->swalk = false;
                    }
                }
                d.o_sinst = append(sinst.data(), sinst.size() * 16);
                d.o_tlas = append(wtlas.data(), wtlas.size() * 16);
                {   // the instance records as they are, pad[0] carrying the instance's class (hit_push<true> reads it from the staged copy)
                    std::vector<spt_instance> inst_copy(s.instances, s.instances + s.n_instances);
                    for (uint32_t i = 0; i < s.n_instances; ++i) { const uint32_t c = cls[i]; std::memcpy(&inst_copy[i].pad[0], &c, 4); }
                    d.o_inst = append(inst_copy.data(), (size_t)s.n_instances * sizeof(spt_instance));
                }
                d.o_mesh = append(mesh_rec.data(), mesh_rec.size() * 16);
                d.o_sph = append(s.spheres, (size_t)s.n_spheres * sizeof(spt_sphere));
                d.o_blas = append(wblas.data(), wblas.size() * 16);
                d.o_tri = append(tri_blob.data(), (size_t)s.n_tris * sizeof(spt_tri_pos));
                d.o_tord = append(tlas_order.data(), tlas_order.size() * sizeof(uint32_t));
            };
            const size_t stack_bytes = (size_t)kLdsStack * 2 * kBlock * sizeof(uint32_t);   // (ref, t0) per LDS level
            assemble(false);
            sc->lds_geo = blob.size() * 16 <= 32u * 1024u && stack_bytes + blob.size() * 16 <= 64u * 1024u;
            if (std::getenv("SPT_NO_LDS_GEO")) sc->lds_geo = false;   // tests: drive small scenes through the large-scene path
            // patches: the streaming walkers of the large-scene path refill lanes whose patch test is over (measured,
            // t_bezier.json 512^2 @ 64 spp: 177 ms LDS-resident nested walkers, 147 ms here; SPT_BEZ_LDS=1 for the former)
            // (not under SPT_REFERENCE_BVH=1: only the exact 2-wide nodes in the reference's visit order reproduce which near
            //  misses of a patch the reference's own culling loses - see bezier_box_margin; the compressed 4-wide form of
            //  the caller's trees, which serves scenes beyond LDS in that mode, can differ from it in such a pixel)
            if (SPT_WITH_BEZIER && own_bvh && std::getenv("SPT_BEZ_LDS") == nullptr) sc->lds_geo = false;
            if (!sc->lds_geo) assemble(true);
            // a handful of primitives: every lane tests all of them (flat.h) instead of walking the trees.  The budget is in
            // triangle tests per ray (an instanced mesh counts once per instance, a sphere as one); SPT_FLAT_BUDGET=0 turns it off.
            {
                uint64_t cost = 0;
                bool ok = sc->lds_geo && own_bvh && s.n_bezier_patches == 0 && s.n_tris < 65536u;   // (the caller's exact trees are walked as they are)
                for (uint32_t i = 0; i < s.n_instances && ok; ++i) {
                    const spt_instance& in = s.instances[i];
                    if (in.prim_type == SPT_PRIM_MESH) cost += s.meshes[in.prim_id].tri_count;
                    else if (in.prim_type == SPT_PRIM_SPHERE) cost += 1;
                    else ok = false;
                }
                const char* fb = std::getenv("SPT_FLAT_BUDGET");
                d.flat = ok && s.n_instances != 0 && cost <= (fb ? (uint64_t)std::atoi(fb) : (uint64_t)kFlatBudget) ? 1u : 0u;
            }
            // fused bounces (k_shade<0, ., kFused>): LDS-resident geometry + the lean simple-scene shade kernel, and
            // the shading tables must fit behind the geometry too (see tab_ld in shading.h)
            {
                bool simple_scene = s.env.width == 0 && s.n_lights > 0;
                for (uint32_t i = 0; i < s.n_materials; ++i) simple_scene = simple_scene && s.materials[i].bxdf == SPT_BXDF_LAMBERT && s.materials[i].recipe == 0;
                for (uint32_t i = 0; i < s.n_lights; ++i) simple_scene = simple_scene && s.lights[i].type <= SPT_LIGHT_SPOT;
                for (uint32_t i = 0; i < s.n_surfaces; ++i) {
                    const spt_surface& sf = s.surfaces[i];
                    float lum = 0.299f * sf.emissive[0] + 0.587f * sf.emissive[1] + 0.114f * sf.emissive[2];
                    simple_scene = simple_scene && sf.inside_medium < 0 && !(lum > 0.0f) && sf.normal_map == 0 && sf.emissive_map == 0;
                }
                const size_t extra = (size_t)s.n_tris * sizeof(spt_tri_attr) + (size_t)s.n_surfaces * sizeof(spt_surface) +
                                     (size_t)s.n_materials * sizeof(spt_material) + (size_t)s.n_lights * sizeof(spt_light) + 64;
                sc->lds_tables = sc->lds_geo && blob.size() * 16 + extra <= 32u * 1024u && stack_bytes + blob.size() * 16 + extra <= 64u * 1024u;
                sc->fused = sc->lds_tables && simple_scene;
                if (sc->lds_tables) {
                    d.o_attr = append(s.tri_attr, (size_t)s.n_tris * sizeof(spt_tri_attr));
                    d.o_surf = append(s.surfaces, (size_t)s.n_surfaces * sizeof(spt_surface));
                    d.o_mat = append(s.materials, (size_t)s.n_materials * sizeof(spt_material));
                    d.o_light = append(s.lights, (size_t)s.n_lights * sizeof(spt_light));
                }
            }
            if (blob.size() > 0x7fffffffull / 16) fail(SPT_ERR_UNSUPPORTED, "scene geometry larger than 32 GiB");
            {   // eye.h: an eye-relative copy per camera position is possible when every box has ONE origin to be relative to
                bool ok = sc->lds_geo && own_bvh && s.n_bezier_patches == 0 && s.n_tris < 65536u && s.n_instances != 0 && std::getenv("SPT_NO_EYE_BLOB") == nullptr;
                std::vector<uint32_t> mesh_uses(s.n_meshes, 0u);
                for (uint32_t i = 0; i < s.n_instances && ok; ++i) {
                    const spt_instance& in = s.instances[i];
                    if (in.prim_type == SPT_PRIM_MESH) ok = ++mesh_uses[in.prim_id] == 1u;
                    else ok = in.prim_type == SPT_PRIM_SPHERE;   // (a sphere's relative centre rides in its instance record: primitives may be shared)
                }
                const size_t eye_bytes = (blob.size() + s.n_tris) * 16;
                ok = ok && eye_bytes <= 36u * 1024u && stack_bytes + eye_bytes <= 64u * 1024u && sc->blas_range.size() == s.n_meshes;
                sc->eye_ok = ok;
                sc->eye_valid = false;
                if (ok) {
                    sc->host_blob = blob;
                    sc->wtlas_f4 = (uint32_t)wtlas.size();
                } else {
                    sc->host_blob.clear();
                }
            }
            sc->geo.upload(blob.data(), blob.size());
            d.geo = sc->geo.as<float4>();
            d.geo_f4 = (uint32_t)blob.size();
            d.lds_f4 = sc->lds_geo ? d.geo_f4 : 0u;   // large scene: global fetches only (see trace.h)
            sc->lds_bytes = stack_bytes + (size_t)d.lds_f4 * 16;
        }
        bool simple = s.env.width == 0 && s.n_lights > 0;
        for (uint32_t i = 0; i < s.n_materials; ++i) simple = simple && s.materials[i].bxdf == SPT_BXDF_LAMBERT;
        for (uint32_t i = 0; i < s.n_lights; ++i) simple = simple && s.lights[i].type <= SPT_LIGHT_SPOT;
        for (uint32_t i = 0; i < s.n_surfaces; ++i) {
            const spt_surface& sf = s.surfaces[i];
            float lum = 0.299f * sf.emissive[0] + 0.587f * sf.emissive[1] + 0.114f * sf.emissive[2];
            simple = simple && sf.inside_medium < 0 && !(lum > 0.0f);
        }
        // image textures: only scenes that sample one per hit pay for the k_shade<2, .> variant
        bool textured = false;
        for (uint32_t i = 0; i < s.n_materials; ++i) textured = textured || s.materials[i].recipe != 0;
        for (uint32_t i = 0; i < s.n_surfaces; ++i) textured = textured || s.surfaces[i].normal_map != 0 || s.surfaces[i].emissive_map != 0;
        for (uint32_t i = 0; i < s.n_materials; ++i) sc->subsurface = sc->subsurface || s.materials[i].substrate == SPT_SUBSTRATE_SUBSURFACE;
        for (uint32_t i = 0; i < s.n_material_recipes; ++i) sc->subsurface = sc->subsurface || s.material_recipes[i].type == SPT_MAT_SUBSURFACE;
        sc->has_probe = sc->subsurface;
        if (s.n_pndfs != 0) {   // glints get a heavy variant of their own: their tree walks would cost k_shade<2> its registers
            sc->subsurface = true;
            sc->has_pndf = true;
            sc->pndfs.upload(s.pndfs, s.n_pndfs);
            sc->pndf_terms.upload(s.pndf_terms, s.n_pndf_terms);
            sc->pndf_nodes.upload(s.pndf_nodes, s.n_pndf_nodes);
            sc->pndf_refs.upload(s.pndf_refs, s.n_pndf_refs);
            sc->pndf_roots.upload(s.pndf_roots, s.n_pndf_roots);
            d.pndfs = sc->pndfs.as<spt_pndf>();
            d.pndf_terms = sc->pndf_terms.as<spt_pndf_term>();
            d.pndf_nodes = sc->pndf_nodes.as<spt_pndf_node>();
            d.pndf_refs = sc->pndf_refs.as<uint32_t>();
            d.pndf_roots = sc->pndf_roots.as<uint32_t>();
        }
        if (sc->subsurface) {
            textured = true;   // k_shade<3> is k_shade<2> + the probe: the texture tables (possibly empty) are uploaded below
            std::vector<float2> cdf(SPT_SS_CDF_SIZE);
            for (uint32_t i = 0; i < SPT_SS_CDF_SIZE; ++i) {
                spt_ss_cdf_entry(i, &cdf[i].x, &cdf[i].y);
                if (i && !(cdf[i].y >= cdf[i - 1].y)) fail(SPT_ERR_UNSUPPORTED, "BSSRDF radius table is not monotonic");   // ss_sample_r bisects it
            }
            sc->ss_cdf.upload(cdf.data(), cdf.size());
            d.ss_cdf = sc->ss_cdf.as<float2>();
        }
        sc->textured = textured;
        if (textured) {
            simple = false;
            // postfix program per texture node (see shading.h "textures"); children precede parents, so a
            // node's program is its children's programs followed by its own op
            struct Prog { std::vector<uint4> code; uint32_t depth; };
            std::vector<uint32_t> chain_pool;
            std::vector<uint4> prog_pool;
            std::vector<uint2> roots(s.n_textures);
            // emit(node, chain): instructions of `node` evaluated under the modifier chain `chain`
            std::function<uint32_t(uint32_t, std::vector<uint32_t>&, std::vector<uint4>&)> emit =
                [&](uint32_t node, std::vector<uint32_t>& chain, std::vector<uint4>& out) -> uint32_t {
                const spt_texture& t = s.textures[node];
                auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
                switch (t.type) {
                case SPT_TEX_SCALAR:
                    out.push_back(make_uint4(TEXOP_SCALAR, bits(t.value[0]), bits(t.value[1]), bits(t.value[2])));
                    return 1u;
                case SPT_TEX_IMAGE: {
                    const uint32_t first = (uint32_t)chain_pool.size();
                    chain_pool.insert(chain_pool.end(), chain.begin(), chain.end());
                    out.push_back(make_uint4(TEXOP_IMAGE, t.image, first, (uint32_t)chain.size()));
                    return 1u;
                }
                case SPT_TEX_SRGB: {
                    const uint32_t d = emit(t.a, chain, out);
                    out.push_back(make_uint4(TEXOP_SRGB, 0, 0, 0));
                    return d;
                }
                case SPT_TEX_MODIFIER: {
                    chain.push_back(node);
                    const uint32_t d = emit(t.a, chain, out);
                    chain.pop_back();
                    return d;
                }
                default: {
                    const uint32_t da = emit(t.a, chain, out);
                    const uint32_t db = emit(t.b, chain, out);
                    out.push_back(make_uint4(TEXOP_ADD + (t.type - SPT_TEX_ADD), 0, 0, 0));
                    return std::max(da, db + 1u);
                }
                }
            };
            for (uint32_t i = 0; i < s.n_textures; ++i) {
                std::vector<uint32_t> chain;
                std::vector<uint4> code;
                const uint32_t depth = emit(i, chain, code);
                if (depth > 4u) fail(SPT_ERR_UNSUPPORTED, "texture expression needs more than 4 pending values");
                if (prog_pool.size() + code.size() > 0x7fffffffull) fail(SPT_ERR_UNSUPPORTED, "texture programs too large");
                roots[i] = make_uint2((uint32_t)prog_pool.size(), (uint32_t)code.size());
                prog_pool.insert(prog_pool.end(), code.begin(), code.end());
            }
            if (chain_pool.empty()) chain_pool.push_back(0u);
            sc->textures.upload(s.textures, s.n_textures);
            sc->tex_prog.upload(prog_pool.data(), prog_pool.size());
            sc->tex_root.upload(roots.data(), roots.size());
            sc->tex_chain.upload(chain_pool.data(), chain_pool.size());
            sc->images.upload(s.images, s.n_images);
            sc->image_levels.upload(s.image_levels, s.n_image_levels);
            sc->texels.upload(s.texels, s.n_texels);
            sc->recipes.upload(s.material_recipes, s.n_material_recipes);
            d.textures = sc->textures.as<float4>();
            d.tex_prog = sc->tex_prog.as<uint4>();
            d.tex_root = sc->tex_root.as<uint2>();
            d.tex_chain = sc->tex_chain.as<uint32_t>();
            d.images = sc->images.as<uint2>();
            d.image_levels = sc->image_levels.as<uint4>();
            d.texels = sc->texels.as<uint32_t>();
            d.recipes = sc->recipes.as<uint4>();
        }
        sc->simple = simple;
        if (s.n_instances) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool finite = true;
            for (uint32_t i = 0; i < s.n_instances; ++i) {
                const double margin = bezier_box_margin(s.instances[i]);   // the screen-space bound and the bounding sphere cull too
                for (int k = 0; k < 3; ++k) {
                    lo[k] = std::min(lo[k], (double)s.instances[i].bmin[k] - margin);
                    hi[k] = std::max(hi[k], (double)s.instances[i].bmax[k] + margin);
                    finite = finite && std::isfinite(s.instances[i].bmin[k]) && std::isfinite(s.instances[i].bmax[k]);
                }
            }
            double r2 = 0;
            for (int k = 0; k < 3; ++k) {
                sc->bs_center[k] = 0.5 * (lo[k] + hi[k]);
                sc->world_lo[k] = lo[k];
                sc->world_hi[k] = hi[k];
                r2 += 0.25 * (hi[k] - lo[k]) * (hi[k] - lo[k]);
            }
            sc->bs_radius = std::sqrt(r2) * 1.001 + 1e-4;
            sc->bs_valid = finite && std::isfinite(sc->bs_radius) && sc->bs_radius < 1e18;
            // Object-space boxes for the per-row spans: the exact bounds of a mesh's vertices / a sphere, padded a little,
            // carried to world space by the instance matrix (a rotated cube's world AABB is far larger than its silhouette).
            // Patches keep their (margin-widened) world box: the clipping test accepts near misses (bezier_box_margin).
            if (sc->bs_valid && s.n_instances <= 256u) {
                sc->hull_corners.resize(s.n_instances);
                for (uint32_t i = 0; i < s.n_instances && !sc->hull_corners.empty(); ++i) {
                    const spt_instance& in = s.instances[i];
                    double olo[3] = {1e300, 1e300, 1e300}, ohi[3] = {-1e300, -1e300, -1e300};
                    bool object_space = true;
                    if (in.prim_type == SPT_PRIM_MESH) {
                        const spt_mesh& m = s.meshes[in.prim_id];
                        for (uint32_t t = m.tri_first; t < m.tri_first + m.tri_count; ++t) {
                            const float* v[3] = {s.tri_pos[t].p0, s.tri_pos[t].p1, s.tri_pos[t].p2};
                            for (int c = 0; c < 3; ++c)
                                for (int k = 0; k < 3; ++k) { olo[k] = std::min(olo[k], (double)v[c][k]); ohi[k] = std::max(ohi[k], (double)v[c][k]); }
                        }
                        if (m.tri_count == 0) { for (int k = 0; k < 3; ++k) { olo[k] = 0; ohi[k] = 0; } }
                    } else if (in.prim_type == SPT_PRIM_SPHERE) {
                        const spt_sphere& sp = s.spheres[in.prim_id];
                        for (int k = 0; k < 3; ++k) { olo[k] = (double)sp.center[k] - std::fabs((double)sp.radius); ohi[k] = (double)sp.center[k] + std::fabs((double)sp.radius); }
                    } else {
                        object_space = false;
                        const double margin = bezier_box_margin(in);
                        for (int k = 0; k < 3; ++k) { olo[k] = (double)in.bmin[k] - margin; ohi[k] = (double)in.bmax[k] + margin; }
                    }
                    double ext = 1e-30;
                    for (int k = 0; k < 3; ++k) ext = std::max(ext, ohi[k] - olo[k]);
                    bool ok = true;
                    for (int c = 0; c < 8; ++c) {
                        double o[3];
                        for (int k = 0; k < 3; ++k) o[k] = ((c >> k) & 1) ? ohi[k] + 1e-4 * ext : olo[k] - 1e-4 * ext;
                        for (int r = 0; r < 3; ++r) {
                            double w = o[r];
                            if (object_space) w = (double)in.fwd[r] * o[0] + (double)in.fwd[3 + r] * o[1] + (double)in.fwd[6 + r] * o[2] + (double)in.fwd[9 + r];
                            sc->hull_corners[i][3 * c + r] = w;
                            ok = ok && std::isfinite(w);
                        }
                    }
                    if (!ok) sc->hull_corners.clear();
                }
            }
        }
        *out = sc;
        return SPT_OK;
    } catch (const AbiError& e) {
        g_error = e.msg;
        delete sc;
        return e.code;
    } catch (const std::exception& e) {
        g_error = std::string("scene_create: ") + e.what();
        delete sc;
        return SPT_ERR_OUT_OF_MEMORY;
    }
}

void spt_scene_destroy(spt_scene* scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    (void)hipStreamSynchronize(scene->stream);
    if (scene->stream_copy) (void)hipStreamSynchronize(scene->stream_copy);
    // every DeviceBuffer member frees itself (a list here used to miss the buffers added later)
    delete scene;
}

// eye.h: the copy of the LDS-resident geometry relative to the camera position `eye` (every value by the f32 operations, in the
// order, in which the walkers of trace.h compute it per ray - this file is compiled with FP contraction off like the kernels),
// uploaded behind the work already queued on the scene's stream.
static void make_eye_blob(spt_scene* sc, const float eye[3]) {
    const DScene& d = sc->d;
    std::vector<float4> eb = sc->host_blob;
    auto sub3 = [](float4& v, const float o[3]) { v.x = v.x - o[0]; v.y = v.y - o[1]; v.z = v.z - o[2]; };
    auto as_u = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    uint32_t n_tris = 0;
    for (uint32_t m = 0; m < d.n_meshes; ++m) {
        const uint32_t range = as_u(eb[d.o_mesh + 2u * m + 1u].w);
        n_tris = std::max(n_tris, (range & 0xffffu) + (range >> 16));
    }
    const uint32_t o_eye = (uint32_t)eb.size();
    eb.resize(eb.size() + n_tris, make_float4(0, 0, 0, 0));
    for (uint32_t i = 0; i < sc->wtlas_f4; ++i) sub3(eb[d.o_tlas + i], eye);                 // TLAS nodes: world space
    for (uint32_t i = 0; i < d.n_instances; ++i) {
        float rec[48];
        std::memcpy(rec, &eb[d.o_inst + 12u * i], sizeof(rec));
        const float* m = rec;                                                                 // spt_instance::inv
        const uint32_t prim_type = as_u(rec[33]), prim_id = as_u(rec[34]);
        float oo[3];                                                                          // xf_point(inv, eye)
        for (int k = 0; k < 3; ++k) oo[k] = ((m[k] * eye[0] + m[3 + k] * eye[1]) + m[6 + k] * eye[2]) + m[9 + k];
        if (prim_type == SPT_PRIM_SPHERE) {                                                   // oc = o' - c, in pad[1..3] of the instance record
            const float4 sp = eb[d.o_sph + prim_id];
            float4& pad = eb[d.o_inst + 12u * i + 11u];
            pad.x = oo[0] - sp.x; pad.y = oo[1] - sp.y; pad.z = oo[2] - sp.z;
            continue;
        }
        sub3(eb[d.o_mesh + 2u * prim_id], oo);                                                // root box of the BLAS
        sub3(eb[d.o_mesh + 2u * prim_id + 1u], oo);
        for (uint32_t n = 4u * sc->blas_range[prim_id].first; n < 4u * sc->blas_range[prim_id].second; ++n) sub3(eb[d.o_blas + n], oo);
        const uint32_t range = as_u(eb[d.o_mesh + 2u * prim_id + 1u].w);
        for (uint32_t k = range & 0xffffu; k < (range & 0xffffu) + (range >> 16); ++k) {
            float4& a = eb[d.o_tri + 3u * k];
            const float4 e1 = eb[d.o_tri + 3u * k + 1u], e2 = eb[d.o_tri + 3u * k + 2u];
            const float sx = oo[0] - a.x, sy = oo[1] - a.y, sz = oo[2] - a.z;                 // s = o' - p0
            const float rx = sy * e1.z - sz * e1.y, ry = sz * e1.x - sx * e1.z, rz = sx * e1.y - sy * e1.x;   // cross(s, e1)
            const float c = (e2.x * rx + e2.y * ry) + e2.z * rz;                              // dot(e2, s x e1)
            a.x = sx; a.y = sy; a.z = sz;
            eb[o_eye + k] = make_float4(rx, ry, rz, c);
        }
    }
    sc->eye_geo.ensure(eb.size() * 16);
    HIP_CHECK(hipMemcpyAsync(sc->eye_geo.p, eb.data(), eb.size() * 16, hipMemcpyHostToDevice, sc->stream));
    HIP_CHECK(hipStreamSynchronize(sc->stream));   // `eb` is pageable; once per camera position
    sc->eye_d = sc->d;
    sc->eye_d.geo = sc->eye_geo.as<float4>();
    sc->eye_d.geo_f4 = sc->eye_d.lds_f4 = (uint32_t)eb.size();
    sc->eye_d.o_eye = o_eye;
    for (int k = 0; k < 3; ++k) { sc->eye_d.tlas_lo[k] = sc->d.tlas_lo[k] - eye[k]; sc->eye_d.tlas_hi[k] = sc->d.tlas_hi[k] - eye[k]; }
    sc->eye_lds_bytes = sc->lds_bytes - (size_t)sc->d.lds_f4 * 16 + eb.size() * 16;
    for (int k = 0; k < 3; ++k) sc->eye_key[k] = eye[k];
    sc->eye_valid = true;
}

spt_status spt_render(const spt_scene* scene_c, const spt_camera* cam, const spt_render_params* params,
                      float* rgb_mean_out, spt_render_stats* stats) {
    if (!scene_c || !cam || !params || !rgb_mean_out) { g_error = "render: null argument"; return SPT_ERR_INVALID_ARG; }
    spt_scene* sc = const_cast<spt_scene*>(scene_c);
    if (sc->fwd) {
        const spt_status st = sc->fwd->render(sc->inner, cam, params, rgb_mean_out, stats);
        if (st != SPT_OK) g_error = sc->fwd->last_error();
        return st;
    }
    std::lock_guard<std::mutex> lock(sc->mu);
    try {
        const spt_render_params& p = *params;
        if (p.width == 0 || p.height == 0 || p.spp == 0) fail(SPT_ERR_INVALID_ARG, "render: width, height and spp must be > 0");
        if (p.max_depth > 255) fail(SPT_ERR_UNSUPPORTED, "render: max_depth > 255");
        if (p.sampler > SPT_SAMPLER_RECURRENCE) fail(SPT_ERR_INVALID_ARG, "render: unknown sampler");
        if (p.sampler == SPT_SAMPLER_JITTERED && (p.division_x == 0 || p.division_y == 0 || p.division_x * p.division_y != p.spp))
            fail(SPT_ERR_INVALID_ARG, "render: jittered sampler needs spp == division_x * division_y");
        const uint32_t shard_count = p.shard_count ? p.shard_count : 1u, strip_rows = p.strip_rows ? p.strip_rows : 1u;
        if (p.shard_index >= shard_count) fail(SPT_ERR_INVALID_ARG, "render: shard_index >= shard_count");
        if ((uint64_t)p.width * p.height > 0xffffffffull) fail(SPT_ERR_UNSUPPORTED, "render: more than 2^32 pixels");
        const uint32_t own_rows = shard_row_count(p);
        const uint64_t own_pix64 = (uint64_t)own_rows * p.width;
        const bool async_out = (p.flags & SPT_RENDER_ASYNC) != 0;
        if (async_out && (stats || (p.flags & (SPT_RENDER_PROFILE | SPT_RENDER_COUNT_VISITS))))
            fail(SPT_ERR_INVALID_ARG, "render: SPT_RENDER_ASYNC returns no stats (stats must be NULL, no PROFILE / COUNT_VISITS)");
        // `stats` belongs to a caller that may have been compiled against an older (shorter) spt_render_stats: fill a
        // local copy and hand back only the bytes the caller says it has (spt_render_params::stats_size, ABI v9)
        spt_render_stats* const stats_out = stats;
        spt_render_stats stats_local;
        std::memset(&stats_local, 0, sizeof stats_local);
        size_t stats_bytes = 0;
        if (stats_out) {
            if (p.stats_size < 8u) fail(SPT_ERR_INVALID_ARG, "render: stats given but params.stats_size is not set (caller built against ABI < 9?)");
            stats_bytes = std::min<size_t>(p.stats_size, sizeof stats_local);
            std::memset(stats_out, 0, stats_bytes);
            stats = &stats_local;
        }
        struct StatsCopy {   // copies on every exit path that got this far, error paths included
            spt_render_stats* dst; const spt_render_stats* src; size_t n;
            ~StatsCopy() { if (dst) std::memcpy(dst, src, n); }
        } stats_copy{stats_out, &stats_local, stats_bytes};
        if (own_pix64 == 0) return SPT_OK;
        if (own_pix64 > 0x7fffffffull) fail(SPT_ERR_UNSUPPORTED, "render: shard larger than 2^31 pixels");
        const uint32_t own_pix = (uint32_t)own_pix64;
        // BoxFilter (src/filter/boxf.rs:11-14): radius_int = ceil(radius - 0.5) neighbour pixels each way
        const float radius = (p.flags & SPT_RENDER_BOX_RADIUS) ? p.filter_radius : 0.5f;
        if (!(radius == radius) || std::fabs(radius) > 64.0f) fail(SPT_ERR_UNSUPPORTED, "render: box filter radius must be finite and at most 64");
        const int32_t R = (int32_t)std::ceil(radius - 0.5f);
        HIP_CHECK(hipSetDevice(sc->device));
        sc->out.ensure((size_t)own_pix * 3 * sizeof(float));

        hipStream_t st = sc->stream;
        const bool profile = (p.flags & SPT_RENDER_PROFILE) != 0;
        // visit counters: only the kernels that fetch their geometry from memory count (an LDS-resident scene is read once
        // per workgroup whatever the rays do)
        const bool count = (p.flags & SPT_RENDER_COUNT_VISITS) != 0 && !sc->lds_geo;
        sc->visits.ensure(12 * sizeof(unsigned long long));
        if (count) HIP_CHECK(hipMemsetAsync(sc->visits.p, 0, 12 * sizeof(unsigned long long), sc->stream));
        // per-kernel event timing needs one stream; so does a scene with an environment (see the bounce loop)
        const bool overlap = !profile && sc->d.env_w == 0u && std::getenv("SPT_NO_OVERLAP") == nullptr;
        const size_t lds = sc->lds_bytes;
        const bool L = sc->lds_geo;
        // primary rays of an LDS-resident scene through the eye-relative copy of its geometry (eye.h), remade when the eye has moved
        const bool use_eye = L && !count && sc->eye_ok && std::getenv("SPT_NO_EYE_BLOB") == nullptr;
        if (use_eye && (!sc->eye_valid || std::memcmp(sc->eye_key, cam->eye, sizeof(sc->eye_key)) != 0)) make_eye_blob(sc, cam->eye);
        // refilling persistent waves for large scenes: on for shadow rays (any-hit walks end at very
        // different times: 10.9 -> 8.8 ms on the 1 M-triangle scene), off for extension rays (28 vs 20 ms)
        const bool dyn_shadow = !L && std::getenv("SPT_NO_DYN_SHADOW") == nullptr;
        // (with the 2-wide nodes the refilling extension kernel was slower, 28 vs 20 ms; with the 4-wide nodes it is
        //  faster, 16.0 vs 17.7 ms on cfg5 - measured - and on by default)
        const bool dyn_extend = !L && std::getenv("SPT_NO_DYN_EXTEND") == nullptr;
        auto env_u32 = [](const char* name, uint32_t dflt) { const char* v = std::getenv(name); return v ? (uint32_t)std::atoi(v) : dflt; };
        const uint32_t kDynBlocks = env_u32("SPT_DYN_BLOCKS", 2048);   // persistent blocks that pull work
        const uint32_t dyn_refill_below = env_u32("SPT_DYN_REFILL", kRefillBelow), dyn_steps = env_u32("SPT_DYN_STEPS", kStepsPerCheck);
        struct Span { int cls; size_t e0; };
        std::vector<Span> spans;
        size_t ev_used = 0;
        auto get_event = [&]() -> hipEvent_t {
            if (ev_used == sc->events.size()) {
                hipEvent_t e;
                HIP_CHECK(hipEventCreate(&e));
                sc->events.push_back(e);
            }
            return sc->events[ev_used++];
        };
        auto begin = [&](int cls) {
            if (!profile) return;
            spans.push_back(Span{cls, ev_used});
            HIP_CHECK(hipEventRecord(get_event(), st));
        };
        auto end = [&]() {
            if (!profile) return;
            HIP_CHECK(hipEventRecord(get_event(), st));
        };
        hipEvent_t ev_total0 = get_event(), ev_total1 = get_event();
        HIP_CHECK(hipEventRecord(ev_total0, st));
        std::vector<uint32_t> h_counts;
        uint64_t seg_closest = 0, seg_shadow = 0, primary_hits = 0, path_vertices = 0, shadow_first = 0, vertices_second = 0;
        uint64_t samples_traced = 0, live_samples = 0;
        // Per-row screen-space spans: every instance's object-space box (8 world-space corners, spt_scene_create) is
        // projected, the convex hull of the 8 image points is the exact silhouette of the box, and row j keeps the pixels
        // from the leftmost to the rightmost hull point within one row of slack above and below, plus one pixel each side.
        // A pixel outside its row's span cannot see any instance with any sample (k_primary's `in_bounds`): the rotated
        // cube of the headline scene fills 19 % of the image, its world-space AABB's rectangle 25 %.
        const int2* row_span_dev = nullptr;
        if (!sc->hull_corners.empty() && sc->d.env_w == 0u && std::getenv("SPT_NO_PIXEL_CULL") == nullptr && std::getenv("SPT_NO_ROW_SPANS") == nullptr) {
            std::vector<double> key = {(double)p.width, (double)p.height, (double)cam->half_cot_half_fov};
            for (int k = 0; k < 3; ++k) { key.push_back(cam->eye[k]); key.push_back(cam->forward[k]); key.push_back(cam->up[k]); key.push_back(cam->right[k]); }
            if (key != sc->span_key) {
                sc->span_key.clear();
                std::vector<int32_t>& sp = sc->span_host;
                sp.assign((size_t)p.height * 2, 0);
                for (uint32_t j = 0; j < p.height; ++j) { sp[2 * j] = (int32_t)p.width; sp[2 * j + 1] = -1; }
                const double W = (double)p.width, H = (double)p.height, aspect = W / H;
                bool ok = true;
                for (const auto& cn : sc->hull_corners) {
                    double px[8], py[8];
                    for (int c = 0; c < 8 && ok; ++c) {
                        double z = 0, xr = 0, yu = 0, n2 = 0;
                        for (int k = 0; k < 3; ++k) {
                            const double v = cn[3 * c + k] - (double)cam->eye[k];
                            z += v * (double)cam->forward[k]; xr += v * (double)cam->right[k]; yu += v * (double)cam->up[k]; n2 += v * v;
                        }
                        if (!(z > 1e-6 * std::sqrt(n2)) || !(z > 0)) { ok = false; break; }   // beside / behind the eye: no finite silhouette
                        const double x = (double)cam->half_cot_half_fov * xr / z, y = (double)cam->half_cot_half_fov * yu / z;
                        px[c] = (x / aspect + 0.5) * W;                 // pixel i covers [i, i + 1)
                        py[c] = H - (y + 0.5) * H;                     // row j covers (j, j + 1]  (k_primary: y = ((H - j - 1) + oy) / H - 0.5)
                        ok = std::isfinite(px[c]) && std::isfinite(py[c]) && std::fabs(px[c]) < 1e9 && std::fabs(py[c]) < 1e9;
                    }
                    if (!ok) break;
                    // convex hull (monotone chain)
                    int idx[8];
                    for (int c = 0; c < 8; ++c) idx[c] = c;
                    std::sort(idx, idx + 8, [&](int a, int b) { return px[a] < px[b] || (px[a] == px[b] && py[a] < py[b]); });
                    int hull[17], hn = 0;
                    auto crs = [&](int o, int a, int b) { return (px[a] - px[o]) * (py[b] - py[o]) - (py[a] - py[o]) * (px[b] - px[o]); };
                    for (int c = 0; c < 8; ++c) { while (hn >= 2 && crs(hull[hn - 2], hull[hn - 1], idx[c]) <= 0) --hn; hull[hn++] = idx[c]; }
                    for (int c = 6, lower = hn + 1; c >= 0; --c) { while (hn >= lower && crs(hull[hn - 2], hull[hn - 1], idx[c]) <= 0) --hn; hull[hn++] = idx[c]; }
                    if (hn > 1) --hn;      // the last point repeats the first
                    double ymin = 1e300, ymax = -1e300;
                    for (int c = 0; c < hn; ++c) { ymin = std::min(ymin, py[hull[c]]); ymax = std::max(ymax, py[hull[c]]); }
                    const int64_t j0 = std::max<int64_t>(0, (int64_t)std::floor(ymin) - 2), j1 = std::min<int64_t>((int64_t)p.height - 1, (int64_t)std::floor(ymax) + 2);
                    for (int64_t j = j0; j <= j1; ++j) {
                        const double ya = (double)j - 1.0, yb = (double)j + 2.0;    // the row's own band (j, j + 1] and one row of slack each way
                        double xmin = 1e300, xmax = -1e300;
                        for (int c = 0; c < hn; ++c) {
                            const int a = hull[c], b = hull[(c + 1) % hn];
                            double xa = px[a], yA = py[a], xb = px[b], yB = py[b];
                            if (yA > yB) { std::swap(xa, xb); std::swap(yA, yB); }
                            if (yB < ya || yA > yb) continue;
                            double x0 = xa, x1 = xb;
                            if (yB > yA) {      // clip the edge to the band
                                const double t0 = std::max(0.0, (ya - yA) / (yB - yA)), t1 = std::min(1.0, (yb - yA) / (yB - yA));
                                x0 = xa + (xb - xa) * t0; x1 = xa + (xb - xa) * t1;
                            }
                            xmin = std::min({xmin, x0, x1}); xmax = std::max({xmax, x0, x1});
                        }
                        if (xmin > xmax) continue;
                        const int32_t lo = (int32_t)std::max(-1.0, std::min(W, std::floor(xmin) - 1.0)), hi = (int32_t)std::max(-1.0, std::min(W, std::floor(xmax) + 1.0));
                        sp[2 * j] = std::min(sp[2 * j], lo);
                        sp[2 * j + 1] = std::max(sp[2 * j + 1], hi);
                    }
                }
                if (ok) {
                    sc->row_span.ensure((size_t)p.height * 2 * sizeof(int32_t));
                    HIP_CHECK(hipMemcpyAsync(sc->row_span.p, sp.data(), (size_t)p.height * 2 * sizeof(int32_t), hipMemcpyHostToDevice, sc->stream));
                    HIP_CHECK(hipStreamSynchronize(sc->stream));   // `sp` is pageable; once per camera
                    sc->span_key = key;
                } else {
                    sp.clear();
                }
            }
            if (!sc->span_key.empty()) row_span_dev = sc->row_span.as<int2>();
        }
        // One window of whole image rows through the wavefront pipeline.  A shard is one window (row_base 0, the
        // strip formula of the ABI); a wide box filter renders bands of consecutive rows (w_count = w_strip = 1).
        // collect: keep every sample's radiance (3 planes [c][sample][pixel] in sc->rad) instead of summing it into
        // the film.  Returns the context the resolve kernels of the caller need.
        auto trace_window = [&](uint32_t row_base, uint32_t rows, uint32_t w_index, uint32_t w_count, uint32_t w_strip, bool collect) -> RenderCtx {
            const uint64_t n_pix64 = (uint64_t)rows * p.width;
            if (n_pix64 > 0x7fffffffull) fail(SPT_ERR_UNSUPPORTED, "render: window larger than 2^31 pixels");
            const uint32_t n_pix = (uint32_t)n_pix64;
            // samples per pass: keep the queues around a few million entries
            uint32_t spp_pass = p.samples_per_pass;
            if (spp_pass == 0) {
                const uint64_t target = 128ull << 20;
                spp_pass = (uint32_t)std::max<uint64_t>(1, target / n_pix);
            }
            spp_pass = std::min(spp_pass, p.spp);
            // queue shards: shard s holds what the primary tiles mapped to it can emit, which also bounds
            // every later generation of that shard
            const uint32_t tiles_x = (p.width + kTile - 1) / kTile, tiles_y = (rows + kTile - 1) / kTile;
            const uint32_t pix_blocks = tiles_x * tiles_y;
            uint32_t max_tiles = 0;
            {
                std::vector<uint32_t> per(kShards, 0u);
                for (uint32_t ty = 0; ty < tiles_y; ++ty)
                    for (uint32_t tx = 0; tx < tiles_x; ++tx) max_tiles = std::max(max_tiles, ++per[(tx + 9u * ty) % kShards]);
            }
            const uint64_t shard_cap64 = (uint64_t)max_tiles * kBlock * spp_pass;
            const uint64_t cap64 = shard_cap64 * kShards;
            if (cap64 > 0x7fffffffull) fail(SPT_ERR_UNSUPPORTED, "render: pass too large (lower samples_per_pass)");
            const size_t cap = (size_t)cap64;
            // collect: every sample of the window is kept (wide box filter), else only the samples of one pass
            const uint64_t rad64 = (uint64_t)n_pix * (collect ? p.spp : spp_pass);

            // the hit queue is binned by BxDF class for the general shade kernels (kernels.h, kClasses): class c lives c * cap
            // entries further.  Memory is what MI355X has (24 B x cap x 8 classes = 26 GB for a 128 M-sample pass)
            const bool fused_here = sc->fused && sc->simple && std::getenv("SPT_NO_FUSED") == nullptr;
            const uint32_t n_classes = (!fused_here && p.max_depth > 1 && cap * (uint64_t)kClasses <= 0xffffffffull && std::getenv("SPT_NO_CLASS_QUEUES") == nullptr) ? kClasses : 1u;
            for (int k = 0; k < 4; ++k) { sc->qa[k].ensure(cap * 16 * (k == 1 ? n_classes : 1u)); sc->qb[k].ensure(cap * 16); }   // (qa[1]: the compact bounce-0 records sit at their hit's index, in every class)
            sc->qa[4].ensure(cap * 8);
            sc->qb[4].ensure(cap * 8);
            sc->hit_f4.ensure(cap * 16 * n_classes);
            sc->hit_inst.ensure(cap * 8 * n_classes);
            // see k_shade<.., kFused>.  Only the lean k_shade<0> variant gains: with the general kernel's 220+ VGPRs
            // the two traversals run at 2 waves / SIMD and cfg4 is faster un-fused (4.30 vs 4.00 Gsamples/s, measured)
            const bool fused = sc->fused && sc->simple && std::getenv("SPT_NO_FUSED") == nullptr;
            if (fused) {
                sc->hit_f4_next.ensure(cap * 16);
                sc->hit_inst_next.ensure(cap * 8);
            }
            for (int k = 0; k < 3; ++k) sc->sh[k].ensure(cap * 16);
            const size_t counts_words = (size_t)(p.max_depth + 1) * Q_KINDS * kShards * 32;
            const size_t counts_bytes = counts_words * sizeof(uint32_t);
            sc->counts.ensure(counts_bytes);
            sc->rad.ensure((size_t)rad64 * 3 * sizeof(float));
            sc->film.ensure((size_t)n_pix * 3 * sizeof(float));
            sc->first_slot.ensure((size_t)n_pix * sizeof(uint32_t));
            sc->slot_bits.ensure((size_t)n_pix * ((spp_pass + 7u) / 8u));

            RenderCtx rc{};
            rc.cam.eye = f3{cam->eye[0], cam->eye[1], cam->eye[2]};
            rc.cam.forward = f3{cam->forward[0], cam->forward[1], cam->forward[2]};
            rc.cam.up = f3{cam->up[0], cam->up[1], cam->up[2]};
            rc.cam.right = f3{cam->right[0], cam->right[1], cam->right[2]};
            rc.cam.half_cot = cam->half_cot_half_fov;
            rc.width = p.width; rc.height = p.height; rc.spp = p.spp; rc.max_depth = p.max_depth;
            rc.sampler = p.sampler; rc.division_x = p.division_x; rc.division_y = p.division_y;
            rc.seed = p.seed;
            rc.shard_index = w_index; rc.shard_count = w_count; rc.strip_rows = w_strip;
            rc.row_base = row_base;
            rc.n_pixels = n_pix;
            rc.rows = rows;
            rc.tiles_x = tiles_x;
            rc.qa = PathQueue{sc->qa[0].as<float4>(), sc->qa[1].as<float4>(), sc->qa[2].as<float4>(), sc->qa[3].as<float4>(), sc->qa[4].as<uint2>()};
            rc.qb = PathQueue{sc->qb[0].as<float4>(), sc->qb[1].as<float4>(), sc->qb[2].as<float4>(), sc->qb[3].as<float4>(), sc->qb[4].as<uint2>()};
            rc.hits = HitQueue{sc->hit_f4.as<float4>(), sc->hit_inst.as<uint2>()};
            rc.hits_next = HitQueue{sc->hit_f4_next.as<float4>(), sc->hit_inst_next.as<uint2>()};
            rc.shadow = ShadowQueue{sc->sh[0].as<float4>(), sc->sh[1].as<float4>(), sc->sh[2].as<float4>()};
            rc.counts = sc->counts.as<uint32_t>();
            rc.shard_cap = (uint32_t)shard_cap64;
            rc.n_classes = n_classes;
            rc.class_cap = (uint32_t)cap;
            rc.rad = sc->rad.as<float>();
            rc.film = sc->film.as<float>();
            rc.first_slot = sc->first_slot.as<uint32_t>();
            rc.aspect = (float)p.width / (float)p.height;   // pt.rs:239
            rc.width_inv = 1.0f / (float)p.width;           // pt.rs:250-251
            rc.height_inv = 1.0f / (float)p.height;
            rc.spp_inv = 1.0f / (float)p.spp;
            {   // pt.rs:253-254, 272-275
                const float spp_sqrt_inv = 1.0f / std::sqrt((float)p.spp);
                rc.aux_dx = rc.aspect * rc.width_inv * spp_sqrt_inv;
                rc.aux_dy = rc.height_inv * spp_sqrt_inv;
            }
            {
                double oc[3], d2 = 0;
                for (int k = 0; k < 3; ++k) { oc[k] = sc->bs_center[k] - (double)cam->eye[k]; d2 += oc[k] * oc[k]; }
                rc.bs_oc = f3{(float)oc[0], (float)oc[1], (float)oc[2]};
                // a little extra slack for the f32 rounding of oc and of the test itself
                rc.bs_c = (float)((d2 - sc->bs_radius * sc->bs_radius) * (1.0 - 1e-5));
                rc.bs_valid = sc->bs_valid ? 1u : 0u;
                // screen-space bound: project the 8 corners of the union of the instance boxes (double precision).
                // A point P is seen through image coordinates (u, v) = ((x / aspect + 0.5) W, (y + 0.5) H) with
                // x = half_cot * (P - eye).right / (P - eye).forward, y likewise with up (k_primary: pt.rs:269-271).
                rc.cull_i0 = 0; rc.cull_i1 = (int32_t)p.width - 1; rc.cull_j0 = 0; rc.cull_j1 = (int32_t)p.height - 1;
                if (sc->bs_valid && std::getenv("SPT_NO_PIXEL_CULL") == nullptr) {
                    double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
                    bool ok = true;
                    const double ext = std::max({sc->world_hi[0] - sc->world_lo[0], sc->world_hi[1] - sc->world_lo[1], sc->world_hi[2] - sc->world_lo[2], 1e-30});
                    for (int c = 0; c < 8 && ok; ++c) {
                        double v[3], z = 0, xr = 0, yu = 0;
                        for (int k = 0; k < 3; ++k) {
                            const double pad = 1e-4 * ext;   // covers the (tiny) padding of the device-side boxes
                            v[k] = (((c >> k) & 1) ? sc->world_hi[k] + pad : sc->world_lo[k] - pad) - (double)cam->eye[k];
                            z += v[k] * (double)cam->forward[k];
                            xr += v[k] * (double)cam->right[k];
                            yu += v[k] * (double)cam->up[k];
                        }
                        if (!(z > 1e-6 * ext)) { ok = false; break; }   // a corner beside / behind the eye: no finite bound
                        const double x = (double)cam->half_cot_half_fov * xr / z, y = (double)cam->half_cot_half_fov * yu / z;
                        const double u = (x / ((double)p.width / (double)p.height) + 0.5) * (double)p.width, w = (y + 0.5) * (double)p.height;
                        umin = std::min(umin, u); umax = std::max(umax, u);
                        vmin = std::min(vmin, w); vmax = std::max(vmax, w);
                    }
                    if (ok && std::isfinite(umin) && std::isfinite(umax) && std::isfinite(vmin) && std::isfinite(vmax)) {
                        // pixel i covers u in [i, i + 1); row j covers v in [H - 1 - j, H - j); one pixel of slack each side
                        const double H = (double)p.height;
                        auto clampi = [](double x, double lo, double hi) { return (int32_t)std::max(lo, std::min(hi, x)); };
                        rc.cull_i0 = clampi(std::floor(umin) - 1.0, -1.0, (double)p.width);
                        rc.cull_i1 = clampi(std::floor(umax) + 1.0, -1.0, (double)p.width);
                        rc.cull_j0 = clampi(std::floor(H - 1.0 - vmax) - 1.0, -1.0, H);
                        rc.cull_j1 = clampi(std::floor(H - 1.0 - vmin) + 2.0, -1.0, H);
                    }
                }
            }

            rc.dyn_refill_below = dyn_refill_below;
            rc.dyn_steps = dyn_steps;
            // rays of a path tracer are short (cfg5: ~4 node + ~2 triangle + ~1 instance records per segment = 2 - 3 rounds):
            // a finished lane that waits several rounds for its wave costs more than the refill check
            // if-if with 4 rounds per check: 87.4 ms; 8 rounds 95.2; while-while (SPT_STREAM_IFIF=0) 100 - 121 ms
            rc.stream_rounds = std::max(1u, std::min(255u, env_u32("SPT_STREAM_ROUNDS", 4u))) | (env_u32("SPT_STREAM_IFIF", 1u) ? 0x100u : 0u);
            rc.stream_refill_below = std::max(1u, std::min(64u, env_u32("SPT_STREAM_REFILL", 40u)));
            rc.visits = sc->visits.as<unsigned long long>();
            rc.debug_normal = (p.flags & SPT_RENDER_DEBUG_NORMAL) ? 1u : 0u;
            rc.row_span = row_span_dev;
            // tiles of this shard that intersect the screen-space bound (all of them with an environment)
            uint32_t active_tiles = pix_blocks;
            uint64_t live_pixels = n_pix;
            if (sc->d.env_w == 0u) {
                active_tiles = 0;
                live_pixels = 0;
                for (uint32_t r = 0; r < rows; ++r) {
                    const uint32_t strip = r / w_strip;
                    const int32_t j = (int32_t)(row_base + (strip * w_count + w_index) * w_strip + (r - strip * w_strip));
                    if (j >= rc.cull_j0 && j <= rc.cull_j1) {
                        int32_t i0 = std::max(rc.cull_i0, 0), i1 = std::min(rc.cull_i1, (int32_t)p.width - 1);
                        if (row_span_dev) { i0 = std::max(i0, sc->span_host[2 * (size_t)j]); i1 = std::min(i1, sc->span_host[2 * (size_t)j + 1]); }
                        live_pixels += (uint64_t)std::max(0, i1 - i0 + 1);
                    }
                }
                for (uint32_t ty = 0; ty < tiles_y; ++ty)
                    for (uint32_t tx = 0; tx < tiles_x; ++tx) {
                        const int32_t i_lo = (int32_t)(tx * kTile), i_hi = (int32_t)std::min(p.width, (tx + 1) * kTile) - 1;
                        bool rows_in = false;
                        for (uint32_t r = ty * kTile; r < std::min(rows, (ty + 1) * kTile) && !rows_in; ++r) {
                            const uint32_t strip = r / w_strip;
                            const int32_t j = (int32_t)(row_base + (strip * w_count + w_index) * w_strip + (r - strip * w_strip));
                            rows_in = j >= rc.cull_j0 && j <= rc.cull_j1 && i_hi >= rc.cull_i0 && i_lo <= rc.cull_i1;
                            if (rows_in && row_span_dev) rows_in = i_hi >= sc->span_host[2 * (size_t)j] && i_lo <= sc->span_host[2 * (size_t)j + 1];
                        }
                        if (rows_in) ++active_tiles;
                    }
            }
            HIP_CHECK(hipMemsetAsync(rc.film, 0, (size_t)n_pix * 3 * sizeof(float), st));
            if (collect) HIP_CHECK(hipMemsetAsync(sc->rad.p, 0, (size_t)rad64 * 3 * sizeof(float), st));   // pixels outside the screen bound write no slots
            bool chunked_any = false;
            // max_depth 0: `while curr_depth < self.max_depth` (pt.rs:48) never runs, every sample is black - environment included.
            // Nothing is traced: the film (and, for a wide box filter, the kept samples) stay at the zeros written above.  (The
            // passes below would mark the hits' radiance slots as owned and no shade launch would ever write them.)
            for (uint32_t s0 = 0; s0 < (p.max_depth == 0u ? 0u : p.spp); s0 += spp_pass) {
                rc.pass_first = s0;
                rc.pass_samples = std::min(spp_pass, p.spp - s0);
                rc.rad_plane = collect ? (size_t)p.spp * n_pix : (size_t)rc.pass_samples * n_pix;
                rc.pack_first = (sc->d.n_instances < (1u << 20) && rc.pass_samples <= 4096u && std::getenv("SPT_NO_PACK_FIRST") == nullptr) ? 1u : 0u;
                rc.rad = sc->rad.as<float>() + (collect ? (size_t)s0 * n_pix : 0);
                begin(SPT_K_OTHER);
                HIP_CHECK(hipMemsetAsync(rc.counts, 0, counts_bytes, st));
                end();
                begin(SPT_K_PRIMARY);
                // sample chunks per tile: aim at ~6144 busy workgroups (24 per CU; 4096 .. 8192 measured within 2 %) given the tiles inside the screen bound
                rc.n_tiles = pix_blocks;
                rc.primary_chunks = 1;
                {
                    uint32_t want = std::min<uint32_t>(64u, (6144u + active_tiles - 1u) / std::max(active_tiles, 1u));
                    if (const char* v = std::getenv("SPT_PRIMARY_CHUNKS")) want = (uint32_t)std::max(1, std::atoi(v));
                    want = std::max(1u, std::min(want, rc.pass_samples));
                    rc.chunk_samples = (rc.pass_samples + want - 1u) / want;
                    rc.chunk_samples = (rc.chunk_samples + 7u) / 8u * 8u;   // slot_bits: a group of 8 samples belongs to one chunk
                    rc.primary_chunks = (rc.pass_samples + rc.chunk_samples - 1u) / rc.chunk_samples;
                }
                const bool stream = sc->swalk && !L;
                // which kernel classes the streaming walker serves (1 primary, 2 shadow, 4 extend).  Measured on cfg5, one box
                // (gpurun_out r2j): extension rays 93.5 ms refilling state machine -> 87.4 ms streaming if-if; primary rays
                // 8.8 -> 12.4 ms and shadow rays 9.8 -> 11.3 ms (coherent / short walks: the state machine's tighter loop wins)
                const uint32_t stream_mask = env_u32("SPT_STREAM_MASK", SPT_WITH_BEZIER ? 6u : 4u);   // (patch scenes: shadow rays too, 214 -> 197 ms on t_catmull.json)
                const bool stream_p = stream && (stream_mask & 1u), stream_s = stream && (stream_mask & 2u), stream_e = stream && (stream_mask & 4u);
                rc.slot_bits = nullptr;
                if (rc.primary_chunks > 1u || collect) {   // collect: every sample owns a slot, which is what the chunked kernel does
                    chunked_any = true;
                    rc.slot_bits = sc->slot_bits.as<uint8_t>();
                    if (stream_p && count) hipLaunchKernelGGL((k_primary_stream<true, true>), dim3(pix_blocks * rc.primary_chunks), dim3(kBlock), lds, st, sc->d, rc);
                    else if (stream_p) hipLaunchKernelGGL((k_primary_stream<true, false>), dim3(pix_blocks * rc.primary_chunks), dim3(kBlock), lds, st, sc->d, rc);
                    else if (L && use_eye) hipLaunchKernelGGL((k_primary<true, true, false, true>), dim3(pix_blocks * rc.primary_chunks), dim3(kBlock), sc->eye_lds_bytes, st, sc->eye_d, rc);
                    else if (L) hipLaunchKernelGGL((k_primary<true, true, false>), dim3(pix_blocks * rc.primary_chunks), dim3(kBlock), lds, st, sc->d, rc);
                    else if (count) hipLaunchKernelGGL((k_primary<false, true, true>), dim3(pix_blocks * rc.primary_chunks), dim3(kBlock), lds, st, sc->d, rc);
                    else hipLaunchKernelGGL((k_primary<false, true, false>), dim3(pix_blocks * rc.primary_chunks), dim3(kBlock), lds, st, sc->d, rc);
                } else {
                    if (stream_p && count) hipLaunchKernelGGL((k_primary_stream<false, true>), dim3(pix_blocks), dim3(kBlock), lds, st, sc->d, rc);
                    else if (stream_p) hipLaunchKernelGGL((k_primary_stream<false, false>), dim3(pix_blocks), dim3(kBlock), lds, st, sc->d, rc);
                    else if (L && use_eye) hipLaunchKernelGGL((k_primary<true, false, false, true>), dim3(pix_blocks), dim3(kBlock), sc->eye_lds_bytes, st, sc->eye_d, rc);
                    else if (L) hipLaunchKernelGGL((k_primary<true, false, false>), dim3(pix_blocks), dim3(kBlock), lds, st, sc->d, rc);
                    else if (count) hipLaunchKernelGGL((k_primary<false, false, true>), dim3(pix_blocks), dim3(kBlock), lds, st, sc->d, rc);
                    else hipLaunchKernelGGL((k_primary<false, false, false>), dim3(pix_blocks), dim3(kBlock), lds, st, sc->d, rc);
                }
                end();
                for (uint32_t b = 0; b < p.max_depth; ++b) {
                    begin(b == 0 ? SPT_K_SHADE_FIRST : SPT_K_SHADE);
                    if (fused) {
                        // shade + shadow + extend of this bounce in one kernel; vertices of bounce b live in
                        // (qa, hits) for even b and in (qb, hits_next) for odd b
                        RenderCtx rb = rc;
                        if (b & 1u) { std::swap(rb.qa, rb.qb); std::swap(rb.hits, rb.hits_next); }
                        // few vertices left after bounce 0 (seen by the previous pass with a counter readback): bounce 1 and
                        // everything after it in ONE launch, each lane following its path to the end (k_shade's kLoop)
                        const bool tail_loop = b == 1 && sc->tail_vertices <= kTailLoopBelow && std::getenv("SPT_NO_TAIL_LOOP") == nullptr;
                        if (b == 0) hipLaunchKernelGGL((k_shade<0, true, true, true, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, rb, b);
                        else if (tail_loop) hipLaunchKernelGGL((k_shade<0, false, true, true, true, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, rb, b);
                        else hipLaunchKernelGGL((k_shade<0, false, true, true, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, rb, b);
                        end();
                        if (tail_loop) break;
                        continue;
                    }
                    // un-fused: the shade stage of bounce b reads the path records its predecessor wrote (ru.qa) through the
                    // hits' source indices and writes the next ones to ru.qb, which the extend stage traces: the two path
                    // queues swap roles every bounce, the hit queue is one buffer
                    RenderCtx ru = rc;
                    if (b & 1u) std::swap(ru.qa, ru.qb);
                    const bool tab = sc->lds_tables && std::getenv("SPT_NO_LDS_TABLES") == nullptr;   // shading tables from LDS (tab_ld)
                    const size_t shade_lds = sc->has_probe ? lds : 0;   // the BSSRDF probe walks the BVH inside k_shade<3 | 5>: traversal stack
#define SPT_LAUNCH_SHADE(FEAT)                                                                                                                 \
        if (tab) {                                                                                                                                 \
            if (b == 0) hipLaunchKernelGGL((k_shade<FEAT, true, false, true, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);  \
            else hipLaunchKernelGGL((k_shade<FEAT, false, false, true, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);        \
        } else {                                                                                                                                   \
            if (b == 0) hipLaunchKernelGGL((k_shade<FEAT, true, false, false, false>), dim3(kPersistentBlocks), dim3(kBlock), shade_lds, st, sc->d, ru, b); \
            else hipLaunchKernelGGL((k_shade<FEAT, false, false, false, false>), dim3(kPersistentBlocks), dim3(kBlock), shade_lds, st, sc->d, ru, b);       \
        }
                    if (sc->simple) { SPT_LAUNCH_SHADE(0) } else if (!sc->textured) { SPT_LAUNCH_SHADE(1) } else if (!sc->subsurface) { SPT_LAUNCH_SHADE(2) }
                    else if (!sc->has_probe) { SPT_LAUNCH_SHADE(4) }          // glints only: no probe, so the geometry's place does not matter
                    else if (tab || !L) { if (sc->has_pndf) { SPT_LAUNCH_SHADE(5) } else { SPT_LAUNCH_SHADE(3) } }
                    else if (sc->has_pndf) {   // geometry in LDS, tables not: the probe still walks the LDS copy (k_shade's kGeoLds)
                        if (b == 0) hipLaunchKernelGGL((k_shade<5, true, false, false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else hipLaunchKernelGGL((k_shade<5, false, false, false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                    } else {
                        if (b == 0) hipLaunchKernelGGL((k_shade<3, true, false, false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else hipLaunchKernelGGL((k_shade<3, false, false, false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                    }
#undef SPT_LAUNCH_SHADE
                    end();
                    // k_shadow(b) and k_extend(b) are independent unless the scene has an environment (then a missing
                    // extension ray adds its term to the same radiance slot the shadow ray of that vertex adds to, and
                    // the reference's order of the two additions has to be kept): without one, the shadow kernel runs on
                    // a side stream next to the extension kernel and is joined before the next stage reads the slots.
                    const bool side = overlap && b + 1 < p.max_depth;
                    hipStream_t ss = side ? sc->stream2 : st;
                    if (side) {
                        HIP_CHECK(hipEventRecord(sc->ev_fork, st));
                        HIP_CHECK(hipStreamWaitEvent(ss, sc->ev_fork, 0));
                    }
                    begin(SPT_K_SHADOW);
                    if (stream_s && count) hipLaunchKernelGGL(k_shadow_stream<true>, dim3(kDynBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    else if (stream_s) hipLaunchKernelGGL(k_shadow_stream<false>, dim3(kDynBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    else if (L && sc->d.flat) hipLaunchKernelGGL((k_shadow<true, false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    else if (L) hipLaunchKernelGGL((k_shadow<true, false>), dim3(kPersistentBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    else if (dyn_shadow && count) hipLaunchKernelGGL(k_shadow_dyn<true>, dim3(kDynBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    else if (dyn_shadow) hipLaunchKernelGGL(k_shadow_dyn<false>, dim3(kDynBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    else if (count) hipLaunchKernelGGL((k_shadow<false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    else hipLaunchKernelGGL((k_shadow<false, false>), dim3(kPersistentBlocks), dim3(kBlock), lds, ss, sc->d, ru, b);
                    end();
                    if (side) HIP_CHECK(hipEventRecord(sc->ev_join, ss));
                    if (b + 1 < p.max_depth) {
                        begin(SPT_K_EXTEND);
                        if (stream_e && count) hipLaunchKernelGGL(k_extend_stream<true>, dim3(kDynBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else if (stream_e) hipLaunchKernelGGL(k_extend_stream<false>, dim3(kDynBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else if (L && sc->d.flat) hipLaunchKernelGGL((k_extend<true, false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else if (L) hipLaunchKernelGGL((k_extend<true, false>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else if (dyn_extend && count) hipLaunchKernelGGL(k_extend_dyn<true>, dim3(kDynBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else if (dyn_extend) hipLaunchKernelGGL(k_extend_dyn<false>, dim3(kDynBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else if (count) hipLaunchKernelGGL((k_extend<false, true>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        else hipLaunchKernelGGL((k_extend<false, false>), dim3(kPersistentBlocks), dim3(kBlock), lds, st, sc->d, ru, b);
                        end();
                    }
                    if (side) HIP_CHECK(hipStreamWaitEvent(st, sc->ev_join, 0));
                }
                if (!collect) {
                    begin(SPT_K_RESOLVE);
                    if (rc.slot_bits != nullptr && env_u32("SPT_RESOLVE_BATCH", 16u) == 32u) hipLaunchKernelGGL(k_resolve_bits<32u>, dim3(pix_blocks), dim3(kBlock), 0, st, rc);
                    else if (rc.slot_bits != nullptr) hipLaunchKernelGGL(k_resolve_bits<16u>, dim3(pix_blocks), dim3(kBlock), 0, st, rc);
                    else hipLaunchKernelGGL(k_resolve, dim3(pix_blocks), dim3(kBlock), 0, st, rc);
                    end();
                }
                if (stats) {
                    h_counts.resize(counts_words);
                    HIP_CHECK(hipMemcpyAsync(h_counts.data(), rc.counts, counts_bytes, hipMemcpyDeviceToHost, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                    seg_closest += (uint64_t)n_pix * rc.pass_samples;
                    auto qsum = [&](uint32_t b, uint32_t q) {
                        uint64_t t = 0;
                        for (uint32_t s = 0; s < kShards; ++s) t += h_counts[((size_t)(b * Q_KINDS + q) * kShards + s) * 32];
                        if (q == Q_HIT)     // the hit queue's other classes (bounce >= 1 of the general pipeline)
                            for (uint32_t c = 1; c < kClasses; ++c)
                                for (uint32_t s = 0; s < kShards; ++s) t += h_counts[((size_t)(b * Q_KINDS + Q_HIT_CLASS1 + c - 1u) * kShards + s) * 32];
                        return t;
                    };
                    if (p.max_depth > 1) sc->tail_vertices = qsum(1, Q_HIT);
                    primary_hits += qsum(0, Q_HIT);
                    shadow_first += qsum(0, Q_SHADOW);
                    if (p.max_depth > 1) vertices_second += qsum(1, Q_HIT);
                    for (uint32_t b = 0; b < p.max_depth; ++b) {
                        path_vertices += qsum(b, Q_HIT);
                        seg_shadow += qsum(b, Q_SHADOW);
                        if (b + 1 < p.max_depth) seg_closest += qsum(b, Q_EXT);
                    }
                }
            }
            samples_traced += (uint64_t)n_pix * p.spp;
            if (chunked_any) live_samples += live_pixels * p.spp;
            return rc;
        };
        if (R <= 0) {
            const RenderCtx rc = trace_window(0, own_rows, p.shard_index, shard_count, strip_rows, false);
            begin(SPT_K_RESOLVE);
            const dim3 grid((own_pix + kBlock - 1) / kBlock);
            if (sc->copy_pending) HIP_CHECK(hipStreamWaitEvent(st, sc->ev_copy_done, 0));   // the previous frame's copy-out reads `out`
            if (radius == 0.5f) hipLaunchKernelGGL(k_finish, dim3((own_pix * 3 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, rc, sc->out.as<float>());
            else hipLaunchKernelGGL(k_finish_box, grid, dim3(kBlock), 0, st, rc, sc->out.as<float>(), radius, R);
            end();
        } else {
            // Film::filter_pixel (film.rs:71-92) reads the samples of (2R+1)^2 pixels: each run of consecutive rows of
            // this shard is rendered as bands of whole rows with R rows of halo, all samples kept, then filtered.
            // The halo rows are traced again by the neighbouring band / rank: samples are a pure function of
            // (seed, pixel, sample), so every copy of a row is the same bits.
            std::vector<uint32_t> own;
            for (uint32_t j = 0; j < p.height; ++j)
                if ((j / strip_rows) % shard_count == p.shard_index) own.push_back(j);
            uint64_t budget = 8ull << 30;   // bytes of kept radiance per band
            if (const char* v = std::getenv("SPT_BOX_BAND_BYTES")) budget = std::max<uint64_t>(1, std::strtoull(v, nullptr, 10));
            const uint64_t per_row = (uint64_t)p.width * p.spp * 3 * sizeof(float);
            const uint64_t fit = std::max<uint64_t>(1, budget / per_row);
            const uint32_t run_max = (uint32_t)std::min<uint64_t>(p.height, fit > 2ull * (uint64_t)R ? fit - 2ull * (uint64_t)R : 1ull);
            for (size_t k = 0; k < own.size();) {
                size_t e = k + 1;
                while (e < own.size() && own[e] == own[e - 1] + 1u && e - k < run_max) ++e;
                const uint32_t j0 = own[k], j1 = own[e - 1] + 1u;
                const uint32_t b0 = j0 >= (uint32_t)R ? j0 - (uint32_t)R : 0u, b1 = (uint32_t)std::min<uint64_t>(p.height, (uint64_t)j1 + (uint64_t)R);
                const RenderCtx rc = trace_window(b0, b1 - b0, 0u, 1u, 1u, true);
                begin(SPT_K_RESOLVE);
                BoxJob job{sc->rad.as<float>(), b0, b1 - b0, j0, j1 - j0, sc->out.as<float>() + k * (size_t)p.width * 3, R, radius};
                if (sc->copy_pending) HIP_CHECK(hipStreamWaitEvent(st, sc->ev_copy_done, 0));
                hipLaunchKernelGGL(k_filter_box, dim3(((j1 - j0) * p.width + kBlock - 1) / kBlock), dim3(kBlock), 0, st, rc, job);
                end();
                k = e;
            }
        }
        HIP_CHECK(hipGetLastError());
        hipStream_t st_out = st;
        if (async_out) {   // the copy-out leaves the compute stream: the next render's kernels run beside it
            HIP_CHECK(hipEventRecord(sc->ev_out_ready, st));
            HIP_CHECK(hipStreamWaitEvent(sc->stream_copy, sc->ev_out_ready, 0));
            st_out = sc->stream_copy;
        }
        {
            hipStream_t st = st_out;
            const size_t strip_bytes = (size_t)strip_rows * p.width * 3 * sizeof(float);
            if (p.out_strip_stride == 0 || p.out_strip_stride == strip_bytes) {
                HIP_CHECK(hipMemcpyAsync(rgb_mean_out, sc->out.p, (size_t)own_pix * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
            } else {
                // strided copy-out: the shard's strips land strip by strip in a larger (full-image) film
                if (p.out_strip_stride < strip_bytes) fail(SPT_ERR_INVALID_ARG, "render: out_strip_stride smaller than a strip");
                const size_t full = own_rows / strip_rows, rest_rows = own_rows - full * strip_rows;
                if (full)
                    HIP_CHECK(hipMemcpy2DAsync(rgb_mean_out, p.out_strip_stride, sc->out.p, strip_bytes, strip_bytes, full, hipMemcpyDeviceToHost, st));
                if (rest_rows)
                    HIP_CHECK(hipMemcpyAsync((char*)rgb_mean_out + full * p.out_strip_stride, (const char*)sc->out.p + full * strip_bytes,
                                             rest_rows * (size_t)p.width * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
            }
        }
        if (async_out) {
            HIP_CHECK(hipEventRecord(sc->ev_copy_done, sc->stream_copy));
            sc->copy_pending = true;
            return SPT_OK;
        }
        unsigned long long h_visits[12] = {};
        if (count) HIP_CHECK(hipMemcpyAsync(h_visits, sc->visits.p, sizeof h_visits, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventRecord(ev_total1, st));
        HIP_CHECK(hipStreamSynchronize(st));
        sc->copy_pending = false;   // (the finish kernels of this render waited for any copy-out still in flight)
        if (stats) {
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 3; ++k) stats->class_visits[c][k] = h_visits[3 * c + k];
            stats->node_visits = h_visits[0] + h_visits[3] + h_visits[6];
            stats->tri_tests = h_visits[1] + h_visits[4] + h_visits[7];
            stats->instance_visits = h_visits[2] + h_visits[5] + h_visits[8];
            stats->node_bytes = stats->node_visits * 64ull;   // wide 2-ary and compressed 4-ary nodes are both 64-byte records
            stats->samples = samples_traced;
            stats->segments_closest = seg_closest;
            stats->segments_shadow = seg_shadow;
            stats->primary_hits = primary_hits;
            stats->path_vertices = path_vertices;
            stats->shadow_first = shadow_first;
            stats->vertices_second = vertices_second;
            stats->live_samples = live_samples;
            float ms = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, ev_total0, ev_total1));
            stats->gpu_ms = ms;
            const bool debug_spans = std::getenv("SPT_DEBUG_SPANS") != nullptr;   // per-launch HIP-event times (profile mode)
            for (auto& sp : spans) {
                float k = 0.0f;
                HIP_CHECK(hipEventElapsedTime(&k, sc->events[sp.e0], sc->events[sp.e0 + 1]));
                stats->kernel_ms[sp.cls] += k;
                stats->kernel_launches[sp.cls] += 1;
                if (debug_spans) std::fprintf(stderr, "[spt] span class %d: %.3f ms\n", sp.cls, k);
            }
        }
        return SPT_OK;
    } catch (const AbiError& e) {
        g_error = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        g_error = std::string("render: ") + e.what();
        return SPT_ERR_OUT_OF_MEMORY;
    }
}

spt_status spt_render_wait(const spt_scene* scene_c) {
    if (!scene_c) { g_error = "render_wait: null argument"; return SPT_ERR_INVALID_ARG; }
    spt_scene* sc = const_cast<spt_scene*>(scene_c);
    if (sc->fwd) {
        const spt_status st = sc->fwd->render_wait(sc->inner);
        if (st != SPT_OK) g_error = sc->fwd->last_error();
        return st;
    }
    std::lock_guard<std::mutex> lock(sc->mu);
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        HIP_CHECK(hipStreamSynchronize(sc->stream));
        HIP_CHECK(hipStreamSynchronize(sc->stream_copy));
        sc->copy_pending = false;
        return SPT_OK;
    } catch (const AbiError& e) {
        g_error = e.msg;
        return e.code;
    }
}

static spt_status trace_common(const spt_scene* scene_c, uint32_t n, const spt_ray* rays, void* out, size_t out_elem, bool closest) {
    if (!scene_c || (n && (!rays || !out))) { g_error = "trace: null argument"; return SPT_ERR_INVALID_ARG; }
    if (n == 0) return SPT_OK;
    spt_scene* sc = const_cast<spt_scene*>(scene_c);
    if (sc->fwd) {
        const spt_status st = closest ? sc->fwd->trace_closest(sc->inner, n, rays, static_cast<spt_hit*>(out))
                                      : sc->fwd->trace_any(sc->inner, n, rays, static_cast<uint8_t*>(out));
        if (st != SPT_OK) g_error = sc->fwd->last_error();
        return st;
    }
    std::lock_guard<std::mutex> lock(sc->mu);
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        sc->trace_in.ensure((size_t)n * sizeof(spt_ray));
        sc->trace_out.ensure((size_t)n * out_elem);
        hipStream_t st = sc->stream;
        HIP_CHECK(hipMemcpyAsync(sc->trace_in.p, rays, (size_t)n * sizeof(spt_ray), hipMemcpyHostToDevice, st));
        const size_t lds = sc->lds_bytes;
        dim3 grid((n + kBlock - 1) / kBlock);
        if (sc->swalk && !sc->lds_geo) {
            if (closest) hipLaunchKernelGGL(k_trace_closest_stream, grid, dim3(kBlock), lds, st, sc->d, n, sc->trace_in.as<spt_ray>(), sc->trace_out.as<spt_hit>());
            else hipLaunchKernelGGL(k_trace_any_stream, grid, dim3(kBlock), lds, st, sc->d, n, sc->trace_in.as<spt_ray>(), sc->trace_out.as<uint8_t>());
        } else if (closest) {
            if (sc->lds_geo) hipLaunchKernelGGL(k_trace_closest<true>, grid, dim3(kBlock), lds, st, sc->d, n, sc->trace_in.as<spt_ray>(), sc->trace_out.as<spt_hit>());
            else hipLaunchKernelGGL(k_trace_closest<false>, grid, dim3(kBlock), lds, st, sc->d, n, sc->trace_in.as<spt_ray>(), sc->trace_out.as<spt_hit>());
        } else {
            if (sc->lds_geo) hipLaunchKernelGGL(k_trace_any<true>, grid, dim3(kBlock), lds, st, sc->d, n, sc->trace_in.as<spt_ray>(), sc->trace_out.as<uint8_t>());
            else hipLaunchKernelGGL(k_trace_any<false>, grid, dim3(kBlock), lds, st, sc->d, n, sc->trace_in.as<spt_ray>(), sc->trace_out.as<uint8_t>());
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, sc->trace_out.p, (size_t)n * out_elem, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return SPT_OK;
    } catch (const AbiError& e) {
        g_error = e.msg;
        return e.code;
    }
}

spt_status spt_alloc_pinned(uint64_t bytes, void** out) {
    if (!out || bytes == 0) { g_error = "alloc_pinned: bad argument"; return SPT_ERR_INVALID_ARG; }
    *out = nullptr;
    if (usable_device_count() <= 0) { g_error = "no HIP device is visible: libspt_hip has no CPU fallback"; return SPT_ERR_NO_DEVICE; }
    hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) { g_error = std::string("hipHostMalloc: ") + hipGetErrorString(e); return SPT_ERR_OUT_OF_MEMORY; }
    return SPT_OK;
}
spt_status spt_pin_host(void* p, uint64_t bytes) {
    if (!p || !bytes) { g_error = "pin_host: null argument"; return SPT_ERR_INVALID_ARG; }
    if (usable_device_count() == 0) { g_error = "no HIP device is visible: libspt_hip has no CPU fallback"; return SPT_ERR_NO_DEVICE; }
    if (hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        g_error = "pin_host: hipHostRegister failed";
        return SPT_ERR_OUT_OF_MEMORY;
    }
    return SPT_OK;
}
void spt_unpin_host(void* p) {
    if (p) (void)hipHostUnregister(p);
}

void spt_free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}

spt_status spt_debug_detmath(int32_t device, uint32_t fn, uint32_t n, const float* a, const float* b, float* out) {
    if (n && (!a || !b || !out)) { g_error = "debug_detmath: null argument"; return SPT_ERR_INVALID_ARG; }
    if (n == 0) return SPT_OK;
    DeviceBuffer da, db, dout;
    try {
        int nd = usable_device_count();
        if (nd <= 0) fail(SPT_ERR_NO_DEVICE, "no HIP device is visible: libspt_hip has no CPU fallback");
        if (device < 0 || device >= nd) fail(SPT_ERR_NO_DEVICE, "device index out of range");
        HIP_CHECK(hipSetDevice(device));
        da.upload(a, n);
        db.upload(b, n);
        dout.alloc((size_t)n * sizeof(float));
        hipLaunchKernelGGL(k_detmath, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, fn, n, da.as<float>(), db.as<float>(), dout.as<float>());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        da.release(); db.release(); dout.release();
        return SPT_OK;
    } catch (const AbiError& e) {
        da.release(); db.release(); dout.release();
        g_error = e.msg;
        return e.code;
    }
}

spt_status spt_debug_bxdf(const spt_scene* scene_c, int32_t device, const spt_material* mt, uint32_t op, uint32_t n, const float* wo,
                          const float* wi_in, const uint64_t* rng_state, float* wi_out, float* f_out, float* pdf_out, int32_t* dir_out) {
    if (!mt || op > 1u || (n && (!wo || !f_out || !pdf_out || (op == 0u ? (!rng_state || !wi_out || !dir_out) : !wi_in)))) {
        g_error = "debug_bxdf: null argument or unknown op";
        return SPT_ERR_INVALID_ARG;
    }
    spt_scene* sc = const_cast<spt_scene*>(scene_c);
    if (sc && sc->fwd) {   // a patch scene lives in the other code object
        const spt_status st = sc->fwd->debug_bxdf(sc->inner, device, mt, op, n, wo, wi_in, rng_state, wi_out, f_out, pdf_out, dir_out);
        if (st != SPT_OK) g_error = sc->fwd->last_error();
        return st;
    }
    if (n == 0) return SPT_OK;
    DeviceBuffer dwo, dwi, drng, dwio, df, dpdf, ddir;
    auto release = [&]() { dwo.release(); dwi.release(); drng.release(); dwio.release(); df.release(); dpdf.release(); ddir.release(); };
    try {
        if (mt->bxdf > SPT_BXDF_PNDF_PLASTIC) fail(SPT_ERR_INVALID_ARG, "debug_bxdf: unknown bxdf");
        if (mt->recipe != 0u) fail(SPT_ERR_INVALID_ARG, "debug_bxdf: the record must be a constant Bxdf (recipe 0)");
        const bool pndf = mt->bxdf == SPT_BXDF_PNDF_CONDUCTOR || mt->bxdf == SPT_BXDF_PNDF_PLASTIC;
        if (pndf && !sc) fail(SPT_ERR_INVALID_ARG, "debug_bxdf: a position-normal-distribution lobe needs the scene that holds its tables");
        if (pndf && (sc->pndfs.bytes == 0 || (size_t)spt_f2u(mt->c1[2]) >= sc->pndfs.bytes / sizeof(spt_pndf))) fail(SPT_ERR_INVALID_ARG, "debug_bxdf: P-NDF index (c1[2]) out of range");
        // the exit point of a Subsurface substrate is a traced probe ray (substrate.rs:231-350): only whole films cover it
        if (mt->substrate == SPT_SUBSTRATE_SUBSURFACE && (mt->bxdf == SPT_BXDF_MICROFACET_PLASTIC || mt->bxdf == SPT_BXDF_SPECULAR_PLASTIC || mt->bxdf == SPT_BXDF_PNDF_PLASTIC))
            fail(SPT_ERR_UNSUPPORTED, "debug_bxdf: the Subsurface substrate samples through a probe ray and has no stand-alone seam");
        if (sc) device = sc->device;
        int nd = usable_device_count();
        if (nd <= 0) fail(SPT_ERR_NO_DEVICE, "no HIP device is visible: libspt_hip has no CPU fallback");
        if (device < 0 || device >= nd) fail(SPT_ERR_NO_DEVICE, "device index out of range");
        HIP_CHECK(hipSetDevice(device));
        DMat m;
        m.bxdf = mt->bxdf;
        m.c0 = f3{mt->c0[0], mt->c0[1], mt->c0[2]};
        m.c1 = f3{mt->c1[0], mt->c1[1], mt->c1[2]};
        m.c2 = f3{mt->c2[0], mt->c2[1], mt->c2[2]};
        m.ax = mt->ax; m.ay = mt->ay; m.ior = mt->ior;
        m.fresnel = mt->fresnel; m.substrate = mt->substrate;
        dwo.upload(wo, (size_t)n * 3);
        if (op == 0u) { drng.upload(rng_state, n); dwio.alloc((size_t)n * 3 * sizeof(float)); ddir.alloc((size_t)n * sizeof(int32_t)); }
        else dwi.upload(wi_in, (size_t)n * 3);
        df.alloc((size_t)n * 3 * sizeof(float));
        dpdf.alloc((size_t)n * sizeof(float));
        const dim3 grid((n + kBlock - 1) / kBlock);
        if (pndf) hipLaunchKernelGGL(k_debug_bxdf<true>, grid, dim3(kBlock), 0, 0, sc->d, m, op, n, dwo.as<float>(), dwi.as<float>(), drng.as<uint64_t>(),
                                     dwio.as<float>(), df.as<float>(), dpdf.as<float>(), ddir.as<int32_t>());
        else hipLaunchKernelGGL(k_debug_bxdf<false>, grid, dim3(kBlock), 0, 0, DScene{}, m, op, n, dwo.as<float>(), dwi.as<float>(), drng.as<uint64_t>(),
                                dwio.as<float>(), df.as<float>(), dpdf.as<float>(), ddir.as<int32_t>());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(f_out, df.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(pdf_out, dpdf.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        if (op == 0u) {
            HIP_CHECK(hipMemcpy(wi_out, dwio.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(dir_out, ddir.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
        release();
        return SPT_OK;
    } catch (const AbiError& e) {
        release();
        g_error = e.msg;
        return e.code;
    }
}

spt_status spt_trace_closest(const spt_scene* scene, uint32_t n, const spt_ray* rays, spt_hit* hits) {
    return trace_common(scene, n, rays, hits, sizeof(spt_hit), true);
}
spt_status spt_trace_any(const spt_scene* scene, uint32_t n, const spt_ray* rays, uint8_t* occluded) {
    return trace_common(scene, n, rays, occluded, 1, false);
}

}  // extern "C"
