// Host-side BVH builder for one frame's triangles (replaces the role of the reference's
// octree, framework/composite.cpp:120-161, as the acceleration structure; the octree's
// effect on WHICH leaves exist and in WHAT order they win ties is carried by the
// primitive ids, not by this tree).
//
// Layout produced (device format, see ptx.hip):
//   node  = 4 x float4 (64 B): both children's boxes + both child references
//           q0 = (lo0.x, lo0.y, lo0.z, hi0.x)  q1 = (hi0.y, hi0.z, lo1.x, lo1.y)
//           q2 = (lo1.z, hi1.x, hi1.y, hi1.z)  q3 = (bits(child0), bits(child1), 0, 0)
//   child reference >= 0: node index (relative to the frame's first node)
//                   <  0: leaf, v = ~ref, first item = v >> 3, item count = (v & 7) + 1
//   order = the item permutation: leaf items [first, first+count) index into it.
#pragma once
#include <cstdint>
#include <vector>

namespace ptx {

struct Aabb {
	float lo[3], hi[3];
};

struct BvhNode {
	float q[16];
};

struct Bvh {
	std::vector<BvhNode> nodes;
	std::vector<int32_t> order;
	int32_t root_ref = -1;  // child reference of the whole tree
	int32_t max_depth = 0;  // deepest leaf (root = 0); the traversal stack needs this many slots
	bool empty = true;
};

#ifndef PTX_BVH_MAX_LEAF
#define PTX_BVH_MAX_LEAF 4
#endif
constexpr int kBvhMaxLeaf = PTX_BVH_MAX_LEAF;
// (a lower cap makes shallower trees, a smaller traversal stack and room for more waves, but worse trees: on the mesh
// scene, whose SAH trees come out 20 deep, a cap of 18 changed nothing and 16 cost the closest-hit kernels 3 %)
#ifndef PTX_BVH_MAX_DEPTH
#define PTX_BVH_MAX_DEPTH 30
#endif
constexpr int kBvhMaxDepth = PTX_BVH_MAX_DEPTH;

// boxes must be finite.  Deterministic (no threading, no RNG).  max_leaf: most items per leaf (1 .. 8).
Bvh build_bvh(std::vector<Aabb> const& boxes, int max_leaf = kBvhMaxLeaf);

}  // namespace ptx
