// Binned-SAH BVH2 builder (host).  See ptx_bvh.hpp for the produced layout.
#include "ptx_bvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace ptx {
namespace {

struct Item {
	Aabb box;
	float c[3];
	int32_t idx;
};

struct Builder {
	std::vector<Item> items;
	Bvh out;
	int max_leaf = kBvhMaxLeaf;

	static Aabb empty_box() {
		float inf = std::numeric_limits<float>::infinity();
		return {{inf, inf, inf}, {-inf, -inf, -inf}};
	}
	static void grow(Aabb& a, Aabb const& b) {
		for (int k = 0; k < 3; ++k) {
			a.lo[k] = std::min(a.lo[k], b.lo[k]);
			a.hi[k] = std::max(a.hi[k], b.hi[k]);
		}
	}
	static float half_area(Aabb const& a) {
		float dx = a.hi[0] - a.lo[0], dy = a.hi[1] - a.lo[1], dz = a.hi[2] - a.lo[2];
		return dx * dy + dy * dz + dz * dx;
	}
	Aabb range_box(int b, int e) const {
		Aabb r = empty_box();
		for (int i = b; i < e; ++i) grow(r, items[i].box);
		return r;
	}
	// Boxes are widened so that a hit computed with the reference's per-primitive arithmetic
	// can never fall outside the node that holds the primitive because of rounding in the
	// node's own (fused-multiply-add) slab test.
	static Aabb padded(Aabb a) {
		for (int k = 0; k < 3; ++k) {
			float m = std::max(std::fabs(a.lo[k]), std::fabs(a.hi[k]));
			float pad = 2e-5f * m + 1e-12f;
			a.lo[k] -= pad;
			a.hi[k] += pad;
		}
		return a;
	}
	static int ceil_log2(int n) {
		int l = 0;
		while ((1 << l) < n) ++l;
		return l;
	}

	int32_t build(int b, int e, int depth) {
		int n = e - b;
		out.max_depth = std::max(out.max_depth, depth);
		if (n <= max_leaf) return ~((b << 3) | (n - 1));

		Aabb cb = empty_box();
		for (int i = b; i < e; ++i)
			for (int k = 0; k < 3; ++k) {
				cb.lo[k] = std::min(cb.lo[k], items[i].c[k]);
				cb.hi[k] = std::max(cb.hi[k], items[i].c[k]);
			}
		int mid = -1;
		bool force_median = depth + ceil_log2((n + max_leaf - 1) / max_leaf) + 2 >= kBvhMaxDepth;
		if (!force_median) {
			constexpr int NB = 16;
			float best_cost = std::numeric_limits<float>::infinity();
			int best_axis = -1, best_bin = -1;
			for (int ax = 0; ax < 3; ++ax) {
				float ext = cb.hi[ax] - cb.lo[ax];
				if (!(ext > 0)) continue;
				Aabb bb[NB];
				int cnt[NB] = {0};
				for (auto& x : bb) x = empty_box();
				float scale = NB / ext;
				for (int i = b; i < e; ++i) {
					int k = std::min(NB - 1, std::max(0, (int) ((items[i].c[ax] - cb.lo[ax]) * scale)));
					++cnt[k];
					grow(bb[k], items[i].box);
				}
				float ra[NB];
				int rc[NB];
				Aabb acc = empty_box();
				int c = 0;
				for (int k = NB - 1; k > 0; --k) {
					grow(acc, bb[k]);
					c += cnt[k];
					ra[k] = c ? half_area(acc) : 0.f;
					rc[k] = c;
				}
				acc = empty_box();
				c = 0;
				for (int k = 0; k < NB - 1; ++k) {
					grow(acc, bb[k]);
					c += cnt[k];
					if (c == 0 || rc[k + 1] == 0) continue;
					float cost = half_area(acc) * c + ra[k + 1] * rc[k + 1];
					if (cost < best_cost) {
						best_cost = cost;
						best_axis = ax;
						best_bin = k;
					}
				}
			}
			if (best_axis >= 0) {
				float ext = cb.hi[best_axis] - cb.lo[best_axis];
				float scale = NB / ext;
				float lo = cb.lo[best_axis];
				auto it = std::stable_partition(items.begin() + b, items.begin() + e, [&](Item const& x) {
					int k = std::min(NB - 1, std::max(0, (int) ((x.c[best_axis] - lo) * scale)));
					return k <= best_bin;
				});
				mid = (int) (it - items.begin());
				if (mid == b || mid == e) mid = -1;
			}
		}
		if (mid < 0) {
			int ax = 0;
			float best = -1;
			for (int k = 0; k < 3; ++k) {
				float ext = cb.hi[k] - cb.lo[k];
				if (ext > best) { best = ext; ax = k; }
			}
			mid = b + n / 2;
			std::stable_sort(items.begin() + b, items.begin() + e, [&](Item const& x, Item const& y) { return x.c[ax] < y.c[ax]; });
		}
		int32_t me = (int32_t) out.nodes.size();
		out.nodes.emplace_back();
		Aabb lb = padded(range_box(b, mid)), rb = padded(range_box(mid, e));
		int32_t l = build(b, mid, depth + 1);
		int32_t r = build(mid, e, depth + 1);
		float* q = out.nodes[me].q;
		q[0] = lb.lo[0]; q[1] = lb.lo[1]; q[2] = lb.lo[2]; q[3] = lb.hi[0];
		q[4] = lb.hi[1]; q[5] = lb.hi[2]; q[6] = rb.lo[0]; q[7] = rb.lo[1];
		q[8] = rb.lo[2]; q[9] = rb.hi[0]; q[10] = rb.hi[1]; q[11] = rb.hi[2];
		std::memcpy(&q[12], &l, 4);
		std::memcpy(&q[13], &r, 4);
		q[14] = q[15] = 0;
		return me;
	}
};

}  // namespace

Bvh build_bvh(std::vector<Aabb> const& boxes, int max_leaf) {
	Builder B;
	B.max_leaf = std::min(std::max(max_leaf, 1), 8);
	B.items.resize(boxes.size());
	for (size_t i = 0; i < boxes.size(); ++i) {
		B.items[i].box = boxes[i];
		B.items[i].idx = (int32_t) i;
		for (int k = 0; k < 3; ++k) B.items[i].c[k] = 0.5f * boxes[i].lo[k] + 0.5f * boxes[i].hi[k];
	}
	if (boxes.empty()) return B.out;
	B.out.empty = false;
	B.out.root_ref = B.build(0, (int) boxes.size(), 0);
	B.out.order.resize(boxes.size());
	for (size_t i = 0; i < boxes.size(); ++i) B.out.order[i] = B.items[i].idx;
	return B.out;
}

}  // namespace ptx
