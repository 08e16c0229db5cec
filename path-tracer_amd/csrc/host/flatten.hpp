// Scene graph -> PtxSceneDesc (include/ptx.h).  This is the hand-over between the host tree
// the loader built (Composite/octree, framework/composite.cpp) and the GPU path.
#pragma once
#include <string>
#include <vector>

#include "ptx.h"
#include "scene.hpp"

struct FlatScene {
	FlatScene() = default;
	FlatScene(FlatScene&&) = default;  // desc points into the vectors: movable, not copyable
	FlatScene& operator=(FlatScene&&) = default;
	FlatScene(FlatScene const&) = delete;
	std::vector<PtxFrame> frames;
	std::vector<PtxPrim> prims;
	std::vector<PtxMaterial> materials;
	std::vector<PtxLight> lights;
	// octants of the root and the child entries under them in visiting order; both empty when the root was not split
	std::vector<PtxGuard> guards;
	std::vector<PtxOccurrence> occurrences;
	PtxSceneDesc desc{};
	// per prim: HitPoint::hit_object of the leaf, and its slash-joined path from the root
	std::vector<std::string> prim_names, prim_paths;
	std::vector<std::shared_ptr<Material>> prim_materials;
	// octree statistics of the host tree (composites, leaf entries incl. duplicates, depth)
	int composites = 0, leaf_entries = 0, max_tree_depth = 0;
};

// Walks scene.root depth first with children in std::map order and lists every leaf once,
// at its first encounter: that order is the reference's tie-break order
// (framework/composite.cpp:87-96).  Throws SceneError for trees the GPU path does not take
// (a transformed Composite below another transformed Composite, non-triangle leaves
// inside a mesh).
FlatScene flatten_scene(Scene const& scene);
