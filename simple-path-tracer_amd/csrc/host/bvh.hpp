// Binned-SAH BVH builder producing the flat 32-byte node array of spt_abi.h.
// Role of the reference's BvhAccel::new (src/primitive/bvh.rs:23-141: 16 buckets,
// three axes, leaf size <= 4).  The tree SHAPE is an implementation detail of the
// accelerator (closest hits do not depend on it), so this is an own design:
// centroid-bound binning and the true surface-area heuristic (the reference's
// `surface_area()` returns the box volume, src/core/bbox.rs:95-102).
#pragma once
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../../include/spt_abi.h"
#include "hmath.hpp"

namespace spt_host {

struct BvhBuildResult {
    std::vector<spt_bvh_node> nodes;  // DFS pre-order, root = 0, child indices relative to 0
    std::vector<uint32_t> order;      // order[k] = original item placed at leaf slot k
};

class BvhBuilder {
   public:
    BvhBuilder(const std::vector<Box>& boxes, uint32_t max_leaf, uint32_t n_bins)
        : boxes_(boxes), max_leaf_(max_leaf), n_bins_(n_bins) {}

    BvhBuildResult build() {
        BvhBuildResult r;
        uint32_t n = (uint32_t)boxes_.size();
        r.order.resize(n);
        std::iota(r.order.begin(), r.order.end(), 0u);
        cent_.resize(n);
        for (uint32_t i = 0; i < n; ++i) cent_[i] = boxes_[i].centroid();
        if (n == 0) return r;
        r.nodes.reserve(2 * n / std::max(1u, max_leaf_ / 2) + 8);
        build_node(r, 0, n);
        return r;
    }

   private:
    const std::vector<Box>& boxes_;
    std::vector<V3> cent_;
    uint32_t max_leaf_, n_bins_;

    static void set_box(spt_bvh_node& nd, const Box& b) {
        nd.bmin[0] = b.lo.x; nd.bmin[1] = b.lo.y; nd.bmin[2] = b.lo.z;
        nd.bmax[0] = b.hi.x; nd.bmax[1] = b.hi.y; nd.bmax[2] = b.hi.z;
    }

    uint32_t build_node(BvhBuildResult& r, uint32_t first, uint32_t last) {
        uint32_t idx = (uint32_t)r.nodes.size();
        r.nodes.emplace_back();
        Box bb, cb;
        for (uint32_t k = first; k < last; ++k) {
            bb.grow(boxes_[r.order[k]]);
            cb.grow(cent_[r.order[k]]);
        }
        set_box(r.nodes[idx], bb);
        uint32_t count = last - first;
        auto make_leaf = [&]() {
            r.nodes[idx].a = first;
            r.nodes[idx].b = SPT_LEAF_FLAG | count;
            return idx;
        };
        if (count <= max_leaf_) return make_leaf();

        // pick the best (axis, bin boundary) by SAH over centroid bins
        float best_cost = 3.4e38f;
        int best_axis = -1;
        uint32_t best_split = 0;
        std::vector<Box> bin_box(n_bins_);
        std::vector<uint32_t> bin_cnt(n_bins_);
        std::vector<float> right_area(n_bins_);
        for (int axis = 0; axis < 3; ++axis) {
            float lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 1e-12f)) continue;
            float scale = (float)n_bins_ / ext;
            for (uint32_t b = 0; b < n_bins_; ++b) { bin_box[b] = Box(); bin_cnt[b] = 0; }
            for (uint32_t k = first; k < last; ++k) {
                uint32_t it = r.order[k];
                uint32_t b = std::min(n_bins_ - 1, (uint32_t)std::max(0.0f, (cent_[it][axis] - lo) * scale));
                bin_box[b].grow(boxes_[it]);
                bin_cnt[b]++;
            }
            Box acc;
            for (uint32_t b = n_bins_ - 1; b > 0; --b) {
                acc.grow(bin_box[b]);
                right_area[b] = acc.area();
            }
            Box lacc;
            uint32_t lcnt = 0;
            for (uint32_t b = 1; b < n_bins_; ++b) {
                lacc.grow(bin_box[b - 1]);
                lcnt += bin_cnt[b - 1];
                uint32_t rcnt = count - lcnt;
                if (lcnt == 0 || rcnt == 0) continue;
                float cost = lacc.area() * (float)lcnt + right_area[b] * (float)rcnt;
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = b; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            // all centroids coincide: split the range in half (keeps leaves <= max_leaf)
            mid = first + count / 2;
        } else {
            float lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = (float)n_bins_ / ext;
            auto it = std::stable_partition(r.order.begin() + first, r.order.begin() + last, [&](uint32_t item) {
                uint32_t b = std::min(n_bins_ - 1, (uint32_t)std::max(0.0f, (cent_[item][best_axis] - lo) * scale));
                return b < best_split;
            });
            mid = (uint32_t)(it - r.order.begin());
            if (mid == first || mid == last) mid = first + count / 2;
        }
        uint32_t l = build_node(r, first, mid);
        uint32_t rr = build_node(r, mid, last);
        r.nodes[idx].a = l;
        r.nodes[idx].b = rr;
        return idx;
    }
};

}  // namespace spt_host
