// Builds every scene factory's output with every host build mode under AddressSanitizer and UBSan
// (host code only: g++, no GPU).  tools/host_sanitize.sh compiles and runs it.
#include <cstdio>
#include <string>
#include <vector>
#include "host.hpp"
using namespace rthip;
int main() {
	const char* names[] = {"default", "simple", "glass", "metal", "wallpaper", "sample", "synthetic:5000", "synthetic:70000"};
	for (const char* nm : names) {
		for (int mode = 0; mode < 4; mode++) {
			std::vector<rthip_object> objs; std::string err;
			if (!generate_scene(nm, 0x5EED0001u + mode, &objs, &err)) { printf("gen %s: %s\n", nm, err.c_str()); return 1; }
			BuildOptions o; o.tree_ref = mode == 1; o.tree_gpu = mode == 2; o.tie_reverse = mode == 3;
			FlatScene s;
			if (!build_flat_scene(objs.data(), (int64_t)objs.size(), &s, &err, o)) { printf("build %s: %s\n", nm, err.c_str()); return 1; }
			std::string first;
			int64_t bad = s.gpu.on ? 0 : validate_flat_scene(s, (int64_t)objs.size(), &first);
			// the host halves of the device-side paths: the layout without records, the records of a refit in the tree's order, the build without the reference pass
			{
				FlatScene plan; std::vector<int32_t> slots;
				if (!plan_device_scene(objs.data(), (int64_t)objs.size(), &plan, &slots, &err, o)) { printf("plan %s: %s\n", nm, err.c_str()); return 1; }
				int64_t hittable = -1;
				if (!check_scene_objects(objs.data(), (int64_t)objs.size(), &err, &hittable)) { printf("check %s: %s\n", nm, err.c_str()); return 1; }
				if (plan.gpu.on && (int64_t)slots.size() != hittable) { printf("plan %s: %zu slots\n", nm, slots.size()); return 1; }
				std::vector<int32_t> ids;
				for (const Prim<float>& p : s.prims32) ids.push_back(p.id);
				FlatScene again;
				if (!refit_flat_scene(objs.data(), (int64_t)objs.size(), ids.data(), (int64_t)ids.size(), &again, &err, o)) { printf("refit %s: %s\n", nm, err.c_str()); return 1; }
				for (size_t i = 0; i < ids.size(); i++)
					if (again.prims32[i].id != s.prims32[i].id || again.prims32[i].rank != s.prims32[i].rank || again.prims64[i].v[0] != s.prims64[i].v[0]) { printf("refit %s: primitive %zu differs\n", nm, i); return 1; }
				BuildOptions e = o; e.ranks_elsewhere = true;
				FlatScene bare;
				if (!build_flat_scene(objs.data(), (int64_t)objs.size(), &bare, &err, e) || bare.prims32.size() != s.prims32.size()) { printf("bare build %s: %s\n", nm, err.c_str()); return 1; }
			}
			printf("%s mode %d: %zu nodes %zu prims depth %d gates %zu violations %lld %s checksum %llx\n", nm, mode, s.nodes32.size(), s.prims32.size(), s.depth, s.gates32.size(), (long long)bad, first.c_str(), (unsigned long long)checksum_flat_scene(s));
		}
	}
	return 0;
}
