// host.cpp — seeded scene factories, Camera::new and the BVH build of librthip (host side).
// All arithmetic here is f64 with separately rounded operations (-ffp-contract=off), as in
// the reference's Rust code.
#include "host.hpp"

#include <algorithm>
#include <functional>
#include <future>
#include <map>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace rthip {

// ---------------------------------------------------------------------------------------
// Host-side generator for the scene factories: the PCG RXS-M-XS 32 stream the kernels use,
// on the fixed stream (pixel = 0xFFFFFFFF, sample = 0).  One draw = one 32-bit word.
// ---------------------------------------------------------------------------------------
namespace {
inline uint32_t mix32(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
struct SceneRng {
	uint32_t state, inc;
	explicit SceneRng(uint32_t seed) {
		uint32_t a = mix32(seed + 0x9E3779B9u * (0xFFFFFFFFu + 1u));
		uint32_t b = mix32(a ^ (0x85EBCA6Bu * 1u));
		state = b;
		inc = (mix32(b + a + 0xC2B2AE35u) << 1) | 1u;
	}
	uint32_t next() {
		uint32_t old = state;
		state = old * 747796405u + inc;
		uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
		return (word >> 22u) ^ word;
	}
	double uniform() { return (double)next() * (1.0 / 4294967296.0); }
	double range(double lo, double hi) { return lo + (hi - lo) * uniform(); }  // src/math.rs:13-15
};

rthip_object make_sphere(double x, double y, double z, double r, int mat, double ar, double ag, double ab, double param) {
	rthip_object o;
	std::memset(&o, 0, sizeof(o));
	o.kind = RTHIP_SPHERE; o.material = mat;
	o.center[0] = x; o.center[1] = y; o.center[2] = z;
	o.extent[0] = r;
	o.albedo[0] = ar; o.albedo[1] = ag; o.albedo[2] = ab;
	o.param = param;
	return o;
}
rthip_object make_cube(double x, double y, double z, double hx, double hy, double hz, int mat, double ar, double ag, double ab,
                       double param) {
	rthip_object o;
	std::memset(&o, 0, sizeof(o));
	o.kind = RTHIP_CUBE; o.material = mat;
	o.center[0] = x; o.center[1] = y; o.center[2] = z;
	o.extent[0] = hx; o.extent[1] = hy; o.extent[2] = hz;
	o.albedo[0] = ar; o.albedo[1] = ag; o.albedo[2] = ab;
	o.param = param;
	return o;
}
const int L = RTHIP_LAMBERTIAN, M = RTHIP_METAL, D = RTHIP_DIELECTRIC;

// One grid cell of src/scene.rs:158-200 / src/scenes.rs:254-295.  The draw order is the
// source's: material selector, centre x, centre z, then per branch the albedo components,
// fuzz, and the sphere-or-cube coin.
void grid_cell(SceneRng& rng, double cx_base, double y, double cz_base, double size, double inner,
               std::vector<rthip_object>* spheres, std::vector<rthip_object>* cubes) {
	double random_mat = rng.range(0.0, 1.0);
	double cx = cx_base + 0.9 * rng.range(0.0, 1.0);
	double cz = cz_base + 0.9 * rng.range(0.0, 1.0);
	double dx = cx - 4.0, dy = y - 0.2, dz = cz - 0.0;
	if (!(std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9)) return;
	if (random_mat < 0.35) {  // diffuse: albedo = random * random, emission = the selector itself
		double a0 = rng.range(0.0, 1.0), a1 = rng.range(0.0, 1.0), a2 = rng.range(0.0, 1.0);
		double b0 = rng.range(0.0, 1.0), b1 = rng.range(0.0, 1.0), b2 = rng.range(0.0, 1.0);
		if (rng.range(0.0, 1.0) > 0.5) spheres->push_back(make_sphere(cx, y, cz, size, L, a0 * b0, a1 * b1, a2 * b2, random_mat));
		else cubes->push_back(make_cube(cx, y, cz, size, size, size, L, a0 * b0, a1 * b1, a2 * b2, random_mat));
	} else if (random_mat < 0.85) {  // metal
		double a0 = rng.range(0.5, 1.0), a1 = rng.range(0.5, 1.0), a2 = rng.range(0.5, 1.0);
		double fuzz = rng.range(0.0, 0.3);
		if (rng.range(0.0, 1.0) > 0.5) spheres->push_back(make_sphere(cx, y, cz, size, M, a0, a1, a2, fuzz));
		else cubes->push_back(make_cube(cx, y, cz, size, size, size, M, a0, a1, a2, fuzz));
	} else {  // glass: an outer shell and an inverted inner one
		if (rng.range(0.0, 1.0) > 0.5) {
			spheres->push_back(make_sphere(cx, y, cz, size, D, 0, 0, 0, 1.5));
			spheres->push_back(make_sphere(cx, y, cz, inner, D, 0, 0, 0, 1.5));
		} else {
			cubes->push_back(make_cube(cx, y, cz, size, size, size, D, 0, 0, 0, 1.5));
			cubes->push_back(make_cube(cx, y, cz, inner, inner, inner, D, 0, 0, 0, 1.5));
		}
	}
}

// Scene::create_sample_scene, src/scene.rs:122-205: one object list in insertion order.
void scene_default(uint32_t seed, std::vector<rthip_object>* out) {
	out->push_back(make_cube(0.0, -1000.0, 0.0, 1000.0, 1000.0, 1000.0, L, 0.5, 0.5, 0.5, 0.0));
	out->push_back(make_sphere(0.0, 1.0, 0.0, 1.0, D, 0, 0, 0, 1.5));
	out->push_back(make_sphere(0.0, 1.0, 0.0, -0.98, D, 0, 0, 0, 1.5));
	out->push_back(make_sphere(4.0, 1.0, 0.0, 1.0, L, 0.4, 0.2, 0.1, 3.0));
	out->push_back(make_sphere(-4.0, 1.0, 0.0, 1.0, M, 0.7, 0.6, 0.5, 0.0));
	out->push_back(make_sphere(10000.0, 5000.0, 10000.0, 7500.0, L, 0.8, 0.4, 0.2, 60.0));
	out->push_back(make_cube(-4.0, 0.5, 2.5, 0.8, 0.8, 0.8, L, 0.4, 0.2, 0.1, 3.0));
	SceneRng rng(seed);
	for (int a = -11; a < 11; a++)
		for (int b = -11; b < 11; b++) grid_cell(rng, (double)a, 0.2, (double)b, 0.2, -0.19, out, out);
}
// scenes::create_sample_scene, src/scenes.rs:201-300, combined spheres-then-cubes
// (combine_spheres_and_cubes, src/scenes.rs:5-14): the "many spheres and cubes" scene.
void scene_sample(uint32_t seed, std::vector<rthip_object>* out) {
	std::vector<rthip_object> spheres, cubes;
	cubes.push_back(make_cube(0.0, -100000.0, 0.0, 100000.0, 100000.0, 100000.0, L, 0.5, 0.5, 0.5, 0.0));
	spheres.push_back(make_sphere(0.0, 1.0, 0.0, 1.0, D, 0, 0, 0, 1.5));
	spheres.push_back(make_sphere(0.0, 1.0, 0.0, -0.98, D, 0, 0, 0, 1.5));
	spheres.push_back(make_sphere(4.0, 1.0, 0.0, 1.0, L, 0.4, 0.2, 0.1, 3.0));
	spheres.push_back(make_sphere(-4.0, 1.0, 0.0, 1.0, M, 0.7, 0.6, 0.5, 0.0));
	spheres.push_back(make_sphere(10000.0, 5000.0, 10000.0, 7500.0, L, 0.8, 0.4, 0.2, 15.0));
	cubes.push_back(make_cube(-4.0, 0.5, 2.5, 0.8, 0.8, 0.8, L, 0.4, 0.2, 0.1, 3.0));
	SceneRng rng(seed);
	for (int a = -6; a < 6; a++)
		for (int b = -6; b < 6; b++) grid_cell(rng, 2.0 * (double)a, 0.4, 2.0 * (double)b, 0.4, -0.38, &spheres, &cubes);
	out->insert(out->end(), spheres.begin(), spheres.end());
	out->insert(out->end(), cubes.begin(), cubes.end());
}
void push_sun_and_ground(std::vector<rthip_object>* spheres, std::vector<rthip_object>* cubes) {
	spheres->push_back(make_sphere(10000.0, 5000.0, 10000.0, 5000.0, L, 0.8, 0.4, 0.2, 15.0));
	cubes->push_back(make_cube(0.0, -100002.0, 0.0, 100000.0, 100000.0, 100000.0, L, 0.5, 0.5, 0.5, 0.0));
}
void scene_simple(std::vector<rthip_object>* out) {  // src/scenes.rs:17-75
	std::vector<rthip_object> s, c;
	s.push_back(make_sphere(0.0, 1.0, -2.0, 0.5, L, 1.0, 0.0, 0.0, 0.5));
	s.push_back(make_sphere(1.0, 0.5, -3.0, 0.8, M, 0.75, 0.75, 0.75, 0.05));
	s.push_back(make_sphere(-1.0, -0.5, -4.0, 1.0, M, 0.75, 0.75, 0.75, 0.1));
	s.push_back(make_sphere(-1.0, 1.0, -4.0, 0.4, M, 0.75, 0.75, 0.75, 0.4));
	push_sun_and_ground(&s, &c);
	out->insert(out->end(), s.begin(), s.end());
	out->insert(out->end(), c.begin(), c.end());
}
void scene_glass(std::vector<rthip_object>* out) {  // src/scenes.rs:78-121
	std::vector<rthip_object> s, c;
	for (int i = 0; i < 5; i++) {
		s.push_back(make_sphere((double)i - 5.0, -1.0, -3.0, 0.5, D, 0, 0, 0, 1.5));
		s.push_back(make_sphere((double)i - 5.0, -1.0, -3.0, -0.49, D, 0, 0, 0, 1.5));
	}
	for (int i = 0; i < 5; i++) c.push_back(make_cube(1.5 * (double)i - 5.0, -1.0, -6.0, 0.5, 0.5, 0.5, D, 0, 0, 0, 1.5));
	push_sun_and_ground(&s, &c);
	out->insert(out->end(), s.begin(), s.end());
	out->insert(out->end(), c.begin(), c.end());
}
void scene_metal(std::vector<rthip_object>* out) {  // src/scenes.rs:124-170
	std::vector<rthip_object> s, c;
	s.push_back(make_sphere(0.0, 1.0, -2.0, 1.5, L, 1.0, 0.0, 0.0, 0.5));
	s.push_back(make_sphere(10000.0, 10000.0, 10000.0, 2500.0, L, 0.8, 0.4, 0.2, 30.0));
	c.push_back(make_cube(0.0, -100002.0, 0.0, 100000.0, 100000.0, 100000.0, L, 0.5, 0.5, 0.5, 0.0));
	for (int i = 0; i < 10; i++) s.push_back(make_sphere((double)i - 5.0, -1.0, -3.0, 0.5, M, 0.75, 0.75, 0.75, (double)i / 10.0));
	out->insert(out->end(), s.begin(), s.end());
	out->insert(out->end(), c.begin(), c.end());
}
void scene_wallpaper(std::vector<rthip_object>* out) {  // src/scenes.rs:173-198
	out->push_back(make_sphere(0.0, 0.0, -2.0, 1.0, D, 0, 0, 0, 1.5));
	out->push_back(make_sphere(0.0, 0.0, -2.0, -0.98, D, 0, 0, 0, 1.5));
	out->push_back(make_cube(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, L, 0.0, 0.0, 0.0, 0.0));
}
// No counterpart in the reference: n objects in all, a ground slab and a sun plus random
// spheres and cubes, half each, with the default scene's material mix (35 % Lambertian,
// 50 % metal, 15 % glass) spread over a 1000 x 40 x 1000 volume.
void scene_synthetic(uint32_t seed, int64_t n, std::vector<rthip_object>* out) {
	out->reserve((size_t)n);
	out->push_back(make_cube(0.0, -1000.0, 0.0, 1000.0, 1000.0, 1000.0, L, 0.5, 0.5, 0.5, 0.0));
	if (n > 1) out->push_back(make_sphere(10000.0, 5000.0, 10000.0, 7500.0, L, 0.8, 0.4, 0.2, 60.0));
	SceneRng rng(seed);
	while ((int64_t)out->size() < n) {
		double sel = rng.uniform();
		double cx = rng.range(-500.0, 500.0), cy = rng.range(0.2, 40.0), cz = rng.range(-500.0, 500.0);
		double size = rng.range(0.1, 0.4);
		bool sphere = rng.uniform() > 0.5;
		int mat; double a0, a1, a2, param;
		if (sel < 0.35) { mat = L; a0 = rng.uniform() * rng.uniform(); a1 = rng.uniform() * rng.uniform(); a2 = rng.uniform() * rng.uniform(); param = 0.0; }
		else if (sel < 0.85) { mat = M; a0 = rng.range(0.5, 1.0); a1 = rng.range(0.5, 1.0); a2 = rng.range(0.5, 1.0); param = rng.range(0.0, 0.3); }
		else { mat = D; a0 = a1 = a2 = 0.0; param = 1.5; }
		if (sphere) out->push_back(make_sphere(cx, cy, cz, size, mat, a0, a1, a2, param));
		else out->push_back(make_cube(cx, cy, cz, size, size, size, mat, a0, a1, a2, param));
	}
}
}  // namespace

bool generate_scene(const std::string& name, uint32_t seed, std::vector<rthip_object>* out, std::string* err) {
	out->clear();
	if (name == "default") scene_default(seed, out);
	else if (name == "sample") scene_sample(seed, out);
	else if (name == "simple") scene_simple(out);
	else if (name == "glass") scene_glass(out);
	else if (name == "metal") scene_metal(out);
	else if (name == "wallpaper") scene_wallpaper(out);
	else if (name.rfind("synthetic:", 0) == 0) {
		char* end = nullptr;
		long long n = std::strtoll(name.c_str() + 10, &end, 10);
		if (n < 1 || n > kMaxObjects || (end && *end)) { *err = "synthetic:N needs 1 <= N <= 2^25"; return false; }
		scene_synthetic(seed, (int64_t)n, out);
	} else {
		*err = "unknown scene '" + name + "'";
		return false;
	}
	return true;
}

// ---------------------------------------------------------------------------------------
// Camera::new, src/camera.rs:25-69, operation for operation.
// ---------------------------------------------------------------------------------------
namespace {
struct V3 { double x, y, z; };
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 unit(V3 a) { double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return v3(a.x / l, a.y / l, a.z / l); }
inline double to_radians(double deg) { return deg * M_PI / 180.0; }  // src/math.rs:8-10
inline void put(double* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
}  // namespace

CameraFields camera_new(const rthip_camera_desc& d) {
	const V3 world_up = v3(0.0, 1.0, 0.0);
	V3 origin = v3(d.origin[0], d.origin[1], d.origin[2]);
	V3 direction = v3(d.direction[0], d.direction[1], d.direction[2]);
	double theta = to_radians(d.fov);
	double h = std::tan(theta / 2.0);
	double viewport_height = 2.0 * h * d.focus_dist;
	double viewport_width = viewport_height * ((double)d.width / (double)d.height);
	V3 w = -unit(direction);
	V3 u = unit(cross(world_up, w));
	V3 v = cross(w, u);
	V3 viewport_u = u * viewport_width;
	V3 viewport_v = (-v) * viewport_height;
	V3 delta_x = viewport_u / (double)d.width;
	V3 delta_y = viewport_v / (double)d.height;
	V3 upper_left = origin - (w * d.focus_dist) - viewport_u / 2.0 - viewport_v / 2.0;
	V3 pixel00 = upper_left + (delta_x + delta_y) * 0.5;
	double defocus_radius = d.focus_dist * std::tan(to_radians(d.defocus_angle / 2.0));
	CameraFields c;
	put(c.origin, origin); put(c.pixel00, pixel00); put(c.delta_x, delta_x); put(c.delta_y, delta_y);
	put(c.disk_x, u * defocus_radius); put(c.disk_y, v * defocus_radius);
	c.defocus_angle = d.defocus_angle;
	c.width = d.width; c.height = d.height;
	return c;
}
void camera_rotation(double yaw_deg, double pitch_deg, double out[3]) {  // src/camera.rs:95-103
	double pitch = pitch_deg * (M_PI / 180.0), yaw = yaw_deg * (M_PI / 180.0);
	out[0] = std::cos(yaw) * std::cos(pitch);
	out[1] = std::sin(pitch);
	out[2] = std::sin(yaw) * std::cos(pitch);
}

// ---------------------------------------------------------------------------------------
// BVH::new, src/bvh.rs:17-81, built over an index array and written out flat.
// ---------------------------------------------------------------------------------------
namespace {
struct Box { double lo[3], hi[3]; };
inline Box box_union(const Box& a, const Box& b) {
	Box r;
	for (int k = 0; k < 3; k++) { r.lo[k] = std::fmin(a.lo[k], b.lo[k]); r.hi[k] = std::fmax(a.hi[k], b.hi[k]); }
	return r;
}
inline float round_down(double v) { float f = (float)v; return ((double)f > v) ? std::nextafterf(f, -INFINITY) : f; }
inline float round_up(double v) { float f = (float)v; return ((double)f < v) ? std::nextafterf(f, INFINITY) : f; }

inline double box_area(const Box& b) {
	double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
	if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0;
	return 2.0 * (dx * dy + dy * dz + dz * dx);
}
inline bool box_contains(const Box& outer, const Box& inner) {
	for (int k = 0; k < 3; k++)
		if (!(outer.lo[k] <= inner.lo[k] && inner.hi[k] <= outer.hi[k])) return false;
	return true;
}
inline Box empty_box() {
	Box b;
	for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
	return b;
}

// Two trees are involved.
//  * The REFERENCE tree (BVH::new, src/bvh.rs:17-81) is built exactly as the reference does.
//    It is never traversed on the GPU; it fixes what depends on it: the tie rank of every
//    object, the leaf order reported to callers, and the gate boxes of inverted spheres.
//  * The DEVICE tree is what the kernels traverse.  A closest hit under the (t, rank) order
//    does not depend on the shape of the tree that finds it, so the device tree is free to be
//    a better one: a surface-area-heuristic build over the primitives that can be hit at all
//    (RTHIP_TREE=sah, the default), or the reference's own shape (RTHIP_TREE=ref).
struct Builder {
	const rthip_object* objs;
	int64_t n_objs;
	std::vector<int32_t> order;  // object ids, permuted in place by the reference's per-level sorts
	FlatScene* out;
	int max_depth = 0;
	int max_leaf = 4;
	int kind_cap = 0;  // surface-area build: at most this many spheres and this many cubes in a leaf, 0 = no cap
	bool threads_on = true;  // large ranges are built on several threads (same result as on one)
	bool tie_reverse = false;  // equal sort keys end up in reversed order (the other extreme of what sort_unstable_by may do)
	int sweep_max = 48;  // exact sweep over sorted centroids up to this many primitives, 32 bins beyond
	double cost_node = 1.0, cost_prim = 0.4;  // a primitive test against a node visit (two box tests); tuned on MI355X

	struct RefNode { size_t begin, end; int left, right; Box ref; };
	std::vector<RefNode> ref_nodes;
	int ref_root = 0;
	std::vector<int32_t> rank;        // per object id
	std::vector<uint32_t> gate_first, gate_count;  // per object id: index of its first gate box and how many it has (0 = none)

	// The box the reference caches per object (src/shapes/sphere.rs:19, cube.rs:17): centre
	// -/+ radius or half extent AS GIVEN, so it is inverted for negative sizes.
	double ref_min(int32_t id, int axis) const {
		const rthip_object& o = objs[id];
		return o.center[axis] - (o.kind == RTHIP_SPHERE ? o.extent[0] : o.extent[axis]);
	}
	double ref_max(int32_t id, int axis) const {
		const rthip_object& o = objs[id];
		return o.center[axis] + (o.kind == RTHIP_SPHERE ? o.extent[0] : o.extent[axis]);
	}
	Box ref_box(int32_t id) const {
		Box b;
		for (int k = 0; k < 3; k++) { b.lo[k] = ref_min(id, k); b.hi[k] = ref_max(id, k); }
		return b;
	}
	// True bounds of the primitive as each precision stores it.
	Box bounds64(int32_t id) const {
		const rthip_object& o = objs[id];
		Box b;
		for (int k = 0; k < 3; k++) {
			double e = std::fabs(o.kind == RTHIP_SPHERE ? o.extent[0] : o.extent[k]);
			b.lo[k] = o.center[k] - e; b.hi[k] = o.center[k] + e;
		}
		return b;
	}
	Box bounds32(int32_t id) const {
		const rthip_object& o = objs[id];
		Box b;
		for (int k = 0; k < 3; k++) {
			if (o.kind == RTHIP_SPHERE) {
				double c = (double)(float)o.center[k], r = std::fabs((double)(float)o.extent[0]);
				b.lo[k] = c - r; b.hi[k] = c + r;
			} else {
				double lo = (double)(float)(o.center[k] - o.extent[k]), hi = (double)(float)(o.center[k] + o.extent[k]);
				b.lo[k] = std::fmin(lo, hi); b.hi[k] = std::fmax(lo, hi);
			}
		}
		return b;
	}
	// A cube with a negative half extent can never be hit (src/shapes/cube.rs:29-37: t1 > t2 on
	// that axis); it shapes the reference tree but is left out of the device tree.
	bool can_be_hit(int32_t id) const {
		const rthip_object& o = objs[id];
		if (o.kind == RTHIP_SPHERE) return true;
		return o.extent[0] >= 0.0 && o.extent[1] >= 0.0 && o.extent[2] >= 0.0;
	}

	// ---- reference tree: src/bvh.rs:17-81 ------------------------------------------------
	// Node count of the reference's tree over `len` objects: fixed by len alone (split at len / 2,
	// leaves of fewer than six), so every subtree knows its slice of ref_nodes in advance and the
	// two halves of a large range can be built by different threads.
	std::map<size_t, size_t> ref_count_memo;
	size_t ref_count(size_t len) {
		if (len < 6) return 1;
		auto it = ref_count_memo.find(len);
		if (it != ref_count_memo.end()) return it->second;
		size_t v = 1 + ref_count(len / 2) + ref_count(len - len / 2);
		ref_count_memo[len] = v;
		return v;
	}
	size_t ref_count_ro(size_t len) const { return len < 6 ? 1 : ref_count_memo.at(len); }
	int ref_build(size_t begin, size_t end) {
		ref_count(end - begin);  // fills the memo for every length that occurs below
		ref_nodes.assign(ref_count_ro(end - begin), RefNode{0, 0, -1, -1, empty_box()});
		return ref_build_at(begin, end, 0, 0);
	}
	int ref_build_at(size_t begin, size_t end, size_t me_index, int depth) {
		static thread_local std::vector<std::pair<double, int32_t>> sort_keys;  // scratch, free again before the recursion
		size_t len = end - begin;
		// :28-47: the axis of largest extent of the cached boxes; the first such axis on ties
		// (sort_unstable_by leaves that open; a three-element sort keeps input order).
		int axis = 0;
		double best_range = 0;
		for (int a = 0; a < 3; a++) {
			double bmin = DBL_MAX, bmax = -DBL_MAX;
			for (size_t i = begin; i < end; i++) {
				bmin = std::fmin(bmin, ref_min(order[i], a));
				bmax = std::fmax(bmax, ref_max(order[i], a));
			}
			double range = bmax - bmin;
			if (a == 0 || range > best_range) { axis = a; best_range = range; }
		}
		// :18-26,49: ascending by min+max on that axis.  Equal keys keep their current order
		// (the reference's unstable sort leaves it unspecified).
		// (keys are gathered once per range: the sort then compares adjacent doubles instead of
		// chasing object records, which is what the build of a million objects spends its time on)
		sort_keys.resize(len);
		for (size_t i = 0; i < len; i++) {
			int32_t id = order[begin + (tie_reverse ? len - 1 - i : i)];  // a stable sort of the reversed range reverses the ties
			sort_keys[i] = {ref_min(id, axis) + ref_max(id, axis), id};
		}
		auto by_key = [](const std::pair<double, int32_t>& a, const std::pair<double, int32_t>& b) { return a.first < b.first; };
		if (threads_on && len >= 131072 && depth < 3) {
			// the few largest ranges: eight slices sorted side by side, then merged pairwise (stable
			// throughout, so the result is the one the plain sort gives)
			constexpr size_t kSlices = 8;
			auto* keys = sort_keys.data();
			size_t cut[kSlices + 1];
			for (size_t k = 0; k <= kSlices; k++) cut[k] = len * k / kSlices;
			std::vector<std::future<void>> jobs;
			for (size_t k = 0; k < kSlices; k++) jobs.push_back(std::async(std::launch::async, [=] { std::stable_sort(keys + cut[k], keys + cut[k + 1], by_key); }));
			for (auto& j : jobs) j.get();
			for (size_t width = 1; width < kSlices; width *= 2) {
				jobs.clear();
				for (size_t k = 0; k + width < kSlices + 1 && k < kSlices; k += 2 * width) {
					const size_t a = cut[k], m = cut[std::min(k + width, kSlices)], e = cut[std::min(k + 2 * width, kSlices)];
					jobs.push_back(std::async(std::launch::async, [=] { std::inplace_merge(keys + a, keys + m, keys + e, by_key); }));
				}
				for (auto& j : jobs) j.get();
			}
		} else {
			std::stable_sort(sort_keys.begin(), sort_keys.end(), by_key);
		}
		for (size_t i = 0; i < len; i++) order[begin + i] = sort_keys[i].second;
		int me = (int)me_index;
		ref_nodes[(size_t)me] = RefNode{begin, end, -1, -1, empty_box()};
		if (len < 6) {  // :52-61
			Box ref = ref_box(order[begin]);
			for (size_t i = begin + 1; i < end; i++) ref = box_union(ref, ref_box(order[i]));
			ref_nodes[(size_t)me].ref = ref;
			return me;
		}
		size_t mid = begin + len / 2;  // :62-79: the upper half is the right child
		const size_t left_index = me_index + 1, right_index = left_index + ref_count_ro(mid - begin);
		int l, r;
		if (threads_on && len >= 32768 && depth < 5) {  // disjoint slices of order[] and ref_nodes[]: the halves do not meet
			auto left = std::async(std::launch::async, [=] { return ref_build_at(begin, mid, left_index, depth + 1); });
			r = ref_build_at(mid, end, right_index, depth + 1);
			l = left.get();
		} else {
			l = ref_build_at(begin, mid, left_index, depth + 1);
			r = ref_build_at(mid, end, right_index, depth + 1);
		}
		ref_nodes[(size_t)me].left = l;
		ref_nodes[(size_t)me].right = r;
		ref_nodes[(size_t)me].ref = box_union(ref_nodes[(size_t)l].ref, ref_nodes[(size_t)r].ref);
		return me;
	}
	// Walks the reference tree left to right: tie ranks (the first object of a leaf wins a tie,
	// src/scene.rs:105; the right child wins a tie between siblings, src/bvh.rs:98-104, so
	// leaves rank left to right and objects inside a leaf in reverse), the leaf order, and the
	// gates: the reference reaches an object only if its per-axis test (src/math.rs:356-371)
	// passes on the box of every node above it.  For a box that contains the object's true
	// bounds that is implied by hitting the object; the other boxes become the object's gates.
	// That happens only to spheres of negative radius, whose own cached box is inverted
	// (src/shapes/sphere.rs:13-19), below a split that parted them from their outer shell.
	void ref_annotate(int node, std::vector<int>& chain, int32_t* next_rank) {
		const RefNode& n = ref_nodes[(size_t)node];
		chain.push_back(node);
		if (n.left < 0) {
			int k = (int)(n.end - n.begin);
			for (int j = 0; j < k; j++) {
				int32_t id = order[n.begin + (size_t)j];
				rank[(size_t)id] = *next_rank + (k - 1 - j);
				out->leaf_ids.push_back(id);
				if (objs[id].kind != RTHIP_SPHERE) continue;
				Box tb = bounds64(id);
				uint32_t first = (uint32_t)out->gates64.size(), count = 0;
				for (int anc : chain) {
					const Box& rb = ref_nodes[(size_t)anc].ref;
					if (box_contains(rb, tb)) continue;
					Gate<double> g64;
					Gate<float> g32;
					for (int q = 0; q < 3; q++) {
						g64.box[q] = rb.lo[q]; g64.box[3 + q] = rb.hi[q];
						g32.box[q] = (float)rb.lo[q]; g32.box[3 + q] = (float)rb.hi[q];
					}
					out->gates64.push_back(g64);
					out->gates32.push_back(g32);
					count++;
				}
				if (count) { gate_first[(size_t)id] = first; gate_count[(size_t)id] = count; }
			}
			*next_rank += k;
			out->leaf_sizes.push_back(k);
			out->n_leaves++;
		} else {
			ref_annotate(n.left, chain, next_rank);
			ref_annotate(n.right, chain, next_rank);
		}
		chain.pop_back();
	}

	// ---- device tree ---------------------------------------------------------------------
	struct Sub { int32_t code; Box b64, b32; };

	// Where nodes and primitives are appended.  Normally the scene itself; a thread that builds one
	// half of a large range appends to a list of its own, which is spliced in afterwards in the
	// order a single thread would have produced (node, left subtree, right subtree).
	struct Sink {
		std::vector<Node<double>> n64;
		std::vector<Node<float>> n32;
		std::vector<Prim<double>> p64;
		std::vector<Prim<float>> p32;
		int max_depth = 0;
	};
	static Sink*& local_sink() { static thread_local Sink* s = nullptr; return s; }
	std::vector<Node<double>>& nodes64() { return local_sink() ? local_sink()->n64 : out->nodes64; }
	std::vector<Node<float>>& nodes32() { return local_sink() ? local_sink()->n32 : out->nodes32; }
	std::vector<Prim<double>>& prims64() { return local_sink() ? local_sink()->p64 : out->prims64; }
	std::vector<Prim<float>>& prims32() { return local_sink() ? local_sink()->p32 : out->prims32; }
	int& deepest() { return local_sink() ? local_sink()->max_depth : max_depth; }
	static int32_t rebase(int32_t code, int32_t node_base, int32_t prim_base) {
		if (code >= 0) return code + node_base;
		const uint32_t c = ~(uint32_t)code;
		return (int32_t)~((((c >> 6) + (uint32_t)prim_base) << 6) | (c & 63u));
	}
	// Appends what a finished thread built to the current sink; returns its root in the new numbering.
	Sub splice(Sink& part, Sub root) {
		const int32_t node_base = (int32_t)nodes64().size(), prim_base = (int32_t)prims64().size();
		for (size_t i = 0; i < part.n64.size(); i++)
			for (int c = 0; c < 2; c++) {
				part.n64[i].child[c] = rebase(part.n64[i].child[c], node_base, prim_base);
				part.n32[i].child[c] = part.n64[i].child[c];
			}
		nodes64().insert(nodes64().end(), part.n64.begin(), part.n64.end());
		nodes32().insert(nodes32().end(), part.n32.begin(), part.n32.end());
		prims64().insert(prims64().end(), part.p64.begin(), part.p64.end());
		prims32().insert(prims32().end(), part.p32.begin(), part.p32.end());
		if (part.max_depth > deepest()) deepest() = part.max_depth;
		root.code = rebase(root.code, node_base, prim_base);
		return root;
	}

	void write_node(int32_t me, const Sub& l, const Sub& r) {
		Node<double>& n64 = nodes64()[(size_t)me];
		Node<float>& n32 = nodes32()[(size_t)me];
		const Sub* ch[2] = {&l, &r};
		for (int c = 0; c < 2; c++) {
			for (int k = 0; k < 3; k++) {
				static const int lo_slot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, hi_slot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
				n64.box[6 * c + lo_slot[k]] = ch[c]->b64.lo[k];
				n64.box[6 * c + hi_slot[k]] = ch[c]->b64.hi[k];
				// f32: centre (in the lo slot) and half extent (in the hi slot), see hit_box_centred.  The
				// half extent is taken from the rounded centre and widened by 2^-20 for the rounding of
				// tc -/+ th in the kernel; an empty box gets a negative one and is never hit.
				const double lo = ch[c]->b32.lo[k], hi = ch[c]->b32.hi[k];
				if (lo <= hi) {
					const float centre = (float)(0.5 * lo + 0.5 * hi);
					const double half = std::fmax(hi - (double)centre, (double)centre - lo);
					n32.box[6 * c + lo_slot[k]] = centre;
					n32.box[6 * c + hi_slot[k]] = round_up(half * (1.0 + 0x1p-20) + 1e-37);
				} else {
					n32.box[6 * c + lo_slot[k]] = 0.0f;
					n32.box[6 * c + hi_slot[k]] = -1e30f;
				}
			}
			n64.child[c] = ch[c]->code;
			n32.child[c] = ch[c]->code;
		}
	}
	int32_t new_node() {
		nodes64().emplace_back();
		nodes32().emplace_back();
		return (int32_t)nodes64().size() - 1;
	}
	// Appends the primitives ids[0..k) as one leaf: spheres first, then cubes, so the leaf loop
	// has no per-primitive type branch.
	// Appends the primitive of object `id` in both precisions.
	void emit_prim(int32_t id) {
		const rthip_object& o = objs[id];
		Prim<double> p64;
		Prim<float> p32;
		for (int q = 0; q < 6; q++) { p64.v[q] = 0.0; p32.v[q] = 0.0f; }
		for (int q = 0; q < 3; q++) { p64.v[q] = o.center[q]; p32.v[q] = (float)o.center[q]; }
		if (o.kind == RTHIP_SPHERE) {
			p64.v[3] = o.extent[0];
			p32.v[3] = (float)o.extent[0];
			// the sphere's gates, as integers in the two spare slots
			prim_set_gates(p64, gate_first[(size_t)id], gate_count[(size_t)id]);
			prim_set_gates(p32, gate_first[(size_t)id], gate_count[(size_t)id]);
		} else {
			for (int q = 0; q < 3; q++) {
				p64.v[3 + q] = o.extent[q];
				static const int lo_slot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, hi_slot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
				p32.v[lo_slot[q]] = (float)(o.center[q] - o.extent[q]);
				p32.v[hi_slot[q]] = (float)(o.center[q] + o.extent[q]);
			}
		}
		p64.id = p32.id = id | (o.material << kMatShift) | (o.kind << kKindShift);
		p64.rank = p32.rank = rank[(size_t)id];
		prims64().push_back(p64);
		prims32().push_back(p32);
	}
	Sub emit_leaf(const int32_t* ids, int k) {
		int32_t first = (int32_t)prims64().size();
		Box b64 = empty_box(), b32 = empty_box();
		int ns = 0, nc = 0;
		for (int pass = 0; pass < 2; pass++) {
			for (int j = 0; j < k; j++) {
				int32_t id = ids[j];
				if ((objs[id].kind == RTHIP_SPHERE) != (pass == 0)) continue;
				if (pass == 0) ns++; else nc++;
				emit_prim(id);
				b64 = box_union(b64, bounds64(id));
				b32 = box_union(b32, bounds32(id));
			}
		}
		return Sub{leaf_code(first, ns, nc), b64, b32};
	}

	// Reference-shaped device tree: the reference's nodes and leaves, minus what cannot be hit.
	Sub shape_ref(int node, int depth) {
		const RefNode& n = ref_nodes[(size_t)node];
		if (n.left < 0) {
			std::vector<int32_t> ids;
			for (size_t i = n.begin; i < n.end; i++)
				if (can_be_hit(order[i])) ids.push_back(order[i]);
			return emit_leaf(ids.data(), (int)ids.size());
		}
		int32_t me = new_node();
		Sub l = shape_ref(n.left, depth + 1);
		Sub r = shape_ref(n.right, depth + 1);
		write_node(me, l, r);
		if (depth + 1 > deepest()) deepest() = depth + 1;
		return Sub{me, box_union(l.b64, r.b64), box_union(l.b32, r.b32)};
	}

	// Surface-area-heuristic device tree.
	struct Item { int32_t id; Box b; double c[3]; };
	std::vector<Item> items;

	Sub sah_leaf(size_t begin, size_t end) {
		int32_t ids[8];
		int k = 0;
		for (size_t i = begin; i < end; i++) ids[k++] = items[i].id;
		// inside a leaf keep the reference's order (ascending tie rank reversed = object order is immaterial; sort by id for determinism)
		std::sort(ids, ids + k);
		return emit_leaf(ids, k);
	}
	Sub sah_build(size_t begin, size_t end, int depth) {
		size_t n = end - begin;
		Box bounds = empty_box(), cb = empty_box();
		for (size_t i = begin; i < end; i++) {
			bounds = box_union(bounds, items[i].b);
			for (int k = 0; k < 3; k++) { cb.lo[k] = std::fmin(cb.lo[k], items[i].c[k]); cb.hi[k] = std::fmax(cb.hi[k], items[i].c[k]); }
		}
		if (n <= 1) return sah_leaf(begin, end);
		// keep the recursion within the traversal stack: balanced median splits once a
		// SAH-shaped subtree could run too deep
		int levels_left = kStackMax - 1 - depth;
		int need = 1;
		while (((size_t)max_leaf << need) < n) need++;
		bool force_median = need + 1 >= levels_left;
		double parent_area = box_area(bounds);
		// a leaf may be capped per kind: the leaf loops of a wave run as long as its fullest leaf of each kind
		bool may_be_leaf = n <= (size_t)max_leaf;
		if (may_be_leaf && kind_cap > 0) {
			int ns = 0, nc = 0;
			for (size_t i = begin; i < end; i++) (objs[items[i].id].kind == RTHIP_SPHERE ? ns : nc)++;
			may_be_leaf = ns <= kind_cap && nc <= kind_cap;
		}
		double best_cost = may_be_leaf ? cost_prim * (double)n : INFINITY;
		int best_axis = -1;
		double best_pos = 0;
		size_t best_left = 0;
		if (!force_median && parent_area > 0) {
			constexpr int kBins = 32;
			for (int axis = 0; axis < 3; axis++) {
				double lo = cb.lo[axis], hi = cb.hi[axis];
				if (!(hi > lo)) continue;
				if (n <= (size_t)sweep_max) {
					// exact sweep over the sorted centroids
					std::sort(items.begin() + (long)begin, items.begin() + (long)end, [axis](const Item& a, const Item& b) {
						return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id);
					});
					std::vector<double> right_area(n);
					Box acc = empty_box();
					for (size_t i = n; i-- > 1;) { acc = box_union(acc, items[begin + i].b); right_area[i] = box_area(acc); }
					acc = empty_box();
					for (size_t i = 1; i < n; i++) {
						acc = box_union(acc, items[begin + i - 1].b);
						if (items[begin + i].c[axis] == items[begin + i - 1].c[axis]) continue;  // split between distinct centroids only
						double cost = cost_node + cost_prim * (box_area(acc) * (double)i + right_area[i] * (double)(n - i)) / parent_area;
						if (cost < best_cost) { best_cost = cost; best_axis = axis; best_pos = 0.5 * (items[begin + i].c[axis] + items[begin + i - 1].c[axis]); best_left = i; }
					}
				} else {
					Box bin_box[kBins];
					size_t bin_n[kBins];
					for (int b = 0; b < kBins; b++) { bin_box[b] = empty_box(); bin_n[b] = 0; }
					double scale = (double)kBins / (hi - lo);
					for (size_t i = begin; i < end; i++) {
						int b = (int)((items[i].c[axis] - lo) * scale);
						b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
						bin_box[b] = box_union(bin_box[b], items[i].b);
						bin_n[b]++;
					}
					double right_area[kBins];
					size_t right_n[kBins];
					Box acc = empty_box();
					size_t cnt = 0;
					for (int b = kBins - 1; b >= 1; b--) { acc = box_union(acc, bin_box[b]); cnt += bin_n[b]; right_area[b] = box_area(acc); right_n[b] = cnt; }
					acc = empty_box();
					cnt = 0;
					for (int b = 1; b < kBins; b++) {
						acc = box_union(acc, bin_box[b - 1]);
						cnt += bin_n[b - 1];
						if (cnt == 0 || right_n[b] == 0) continue;
						double cost = cost_node + cost_prim * (box_area(acc) * (double)cnt + right_area[b] * (double)right_n[b]) / parent_area;
						if (cost < best_cost) { best_cost = cost; best_axis = axis; best_pos = lo + (double)b / scale; best_left = cnt; }
					}
				}
			}
		}
		size_t mid;
		if (best_axis >= 0) {
			int axis = best_axis;
			double pos = best_pos;
			auto it = std::partition(items.begin() + (long)begin, items.begin() + (long)end, [axis, pos](const Item& a) { return a.c[axis] < pos; });
			mid = (size_t)(it - items.begin());
			if (mid == begin || mid == end) best_axis = -1;  // numerical corner: fall through to the median
		}
		if (best_axis < 0) {
			if (may_be_leaf) return sah_leaf(begin, end);
			// no useful SAH split (coincident centroids, a forced balanced split, or a leaf too large): median of the widest centroid axis
			int axis = 0;
			for (int k = 1; k < 3; k++)
				if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
			mid = begin + n / 2;
			std::nth_element(items.begin() + (long)begin, items.begin() + (long)mid, items.begin() + (long)end, [axis](const Item& a, const Item& b) {
				return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id);
			});
		}
		(void)best_left;
		int32_t me = new_node();
		Sub l, r;
		if (threads_on && n >= 32768 && depth < 5) {
			// the halves are disjoint slices of items[]: build them side by side.  The left one goes straight
			// into this thread's list, where it follows the node as in a single-threaded build; the right
			// one into a list of its own that is appended when both are done (so an element is copied once per
			// right turn above it, not once per level).
			Sink right_part;
			auto right = std::async(std::launch::async, [&, mid, end, depth] {
				local_sink() = &right_part;
				Sub sub = sah_build(mid, end, depth + 1);
				local_sink() = nullptr;
				return sub;
			});
			l = sah_build(begin, mid, depth + 1);
			r = right.get();
			r = splice(right_part, r);
		} else {
			l = sah_build(begin, mid, depth + 1);
			r = sah_build(mid, end, depth + 1);
		}
		write_node(me, l, r);
		if (depth + 1 > deepest()) deepest() = depth + 1;
		return Sub{me, box_union(l.b64, r.b64), box_union(l.b32, r.b32)};
	}
};
}  // namespace

BuildOptions BuildOptions::from_env() {
	BuildOptions o;
	if (const char* v = std::getenv("RTHIP_TREE")) { o.tree_ref = std::string(v) == "ref"; o.tree_gpu = std::string(v) == "gpu"; }
	if (const char* v = std::getenv("RTHIP_REF_GPU")) o.ref_gpu = std::atoi(v);
	if (const char* v = std::getenv("RTHIP_GPU_LEAF")) o.gpu_leaf = std::atoi(v);
	if (const char* v = std::getenv("RTHIP_SAH_LEAF")) o.sah_leaf = std::atoi(v);
	if (const char* v = std::getenv("RTHIP_SAH_KIND_CAP")) o.sah_kind_cap = std::atoi(v);
	if (const char* v = std::getenv("RTHIP_SAH_CPRIM")) o.sah_cprim = std::atof(v);
	if (const char* v = std::getenv("RTHIP_SAH_SWEEP")) o.sah_sweep = std::atoi(v);
	if (const char* v = std::getenv("RTHIP_BIG_LEAF_MIN")) o.big_leaf_min = std::atoi(v);
	o.no_big_leaf = std::getenv("RTHIP_NO_BIG_LEAF") != nullptr;
	o.no_qnodes = std::getenv("RTHIP_NO_QNODES") != nullptr;
	if (const char* v = std::getenv("RTHIP_BUILD_THREADS")) o.threads = std::atoi(v) != 1;
	o.timing = std::getenv("RTHIP_BUILD_TIMING") != nullptr;
	if (const char* v = std::getenv("RTHIP_TIE_REVERSE")) o.tie_reverse = std::atoi(v) != 0;
	return o;
}

namespace {
// Checks every object (known kind and material, finite fields) and counts those that can be hit
// at all (every sphere; cubes without a negative half extent).  Large scenes on eight threads.
bool check_objects(const rthip_object* objs, int64_t n, std::string* err, int64_t* hittable = nullptr) {
	if (n <= 0) { *err = "empty scene: BVH::new returns None"; return false; }
	if (n > kMaxObjects) { *err = "too many objects"; return false; }
	struct Part { int64_t first_bad = -1, hittable = 0; };
	const auto scan = [objs](int64_t begin, int64_t end) {
		Part p;
		for (int64_t i = begin; i < end; i++) {
			const rthip_object& o = objs[i];
			bool ok = (o.kind == RTHIP_SPHERE || o.kind == RTHIP_CUBE) && o.material >= 0 && o.material <= 2 && std::isfinite(o.param);
			for (int k = 0; k < 3; k++) ok = ok && std::isfinite(o.center[k]) && std::isfinite(o.albedo[k]) && (o.kind == RTHIP_SPHERE ? std::isfinite(o.extent[0]) : std::isfinite(o.extent[k]));
			if (!ok) { p.first_bad = i; return p; }
			p.hittable += (o.kind == RTHIP_SPHERE || (o.extent[0] >= 0.0 && o.extent[1] >= 0.0 && o.extent[2] >= 0.0)) ? 1 : 0;
		}
		return p;
	};
	Part all;
	if (n >= 65536) {
		constexpr int kParts = 8;
		std::vector<std::future<Part>> jobs;
		for (int k = 0; k < kParts; k++) jobs.push_back(std::async(std::launch::async, scan, n * k / kParts, n * (k + 1) / kParts));
		for (auto& j : jobs) {
			const Part p = j.get();
			if (p.first_bad >= 0 && all.first_bad < 0) all.first_bad = p.first_bad;  // parts are in order: the first one reported is the lowest
			all.hittable += p.hittable;
		}
	} else {
		all = scan(0, n);
	}
	if (all.first_bad >= 0) { *err = "object " + std::to_string(all.first_bad) + " has an invalid kind, material or a non-finite field"; return false; }
	if (hittable) *hittable = all.hittable;
	return true;
}
// What every build starts with: the input checked, the materials by object id, and the reference's
// tree (src/bvh.rs:17-81) with what depends on it: tie ranks, leaf order, gates, its depth.
bool reference_pass(const rthip_object* objs, int64_t n, FlatScene* out, std::string* err, const BuildOptions& opt, Builder* bp,
                    const std::function<void(const char*)>& lap) {
	Builder& b = *bp;
	*out = FlatScene();
	if (!check_objects(objs, n, err)) return false;
	b.objs = objs;
	b.n_objs = n;
	b.out = out;
	b.order.resize((size_t)n);
	std::iota(b.order.begin(), b.order.end(), 0);
	b.rank.assign((size_t)n, 0);
	b.gate_first.assign((size_t)n, 0u);
	b.gate_count.assign((size_t)n, 0u);
	out->mats64.resize((size_t)n);
	out->mats32.resize((size_t)n);
	for (int64_t i = 0; i < n; i++) {
		for (int k = 0; k < 3; k++) { out->mats64[(size_t)i].albedo[k] = objs[i].albedo[k]; out->mats32[(size_t)i].albedo[k] = (float)objs[i].albedo[k]; }
		out->mats64[(size_t)i].param = objs[i].param;
		out->mats32[(size_t)i].param = (float)objs[i].param;
	}
	b.threads_on = opt.threads;
	b.tie_reverse = opt.tie_reverse;
	if (opt.ranks_elsewhere && !opt.tree_ref) return true;  // the caller has ranks and gates from the device and rewrites the records
	int ref_root = b.ref_build(0, (size_t)n);
	b.ref_root = ref_root;
	lap("reference tree");
	{
		std::vector<int> chain;
		int32_t next_rank = 0;
		b.ref_annotate(ref_root, chain, &next_rank);
	}
	// reference depth in inner-node levels (reported to callers)
	{
		std::vector<std::pair<int, int>> todo{{ref_root, 0}};
		int d = 1;
		while (!todo.empty()) {
			auto [node, depth] = todo.back();
			todo.pop_back();
			const Builder::RefNode& rn = b.ref_nodes[(size_t)node];
			if (rn.left >= 0) { d = std::max(d, depth + 1); todo.push_back({rn.left, depth + 1}); todo.push_back({rn.right, depth + 1}); }
		}
		out->ref_depth = d;
	}
	lap("ranks, gates, depth");
	return true;
}
// Build option tree_gpu: the primitives laid out for the device build (the leaf of oversized
// primitives first, spheres before cubes; every other primitive behind it, spheres before cubes, in
// object order), the nodes reserved, node 0 written if there is such a leaf, the box of the
// centres.  With `slot_ids` the primitives are not written, only listed (object id per place):
// the caller makes them on the device.  False (and nothing done) for scenes of fewer than two leaves.
bool layout_for_device_build(Builder& b, const BuildOptions& opt, std::vector<int32_t>* slot_ids) {
	FlatScene* out = b.out;
	const rthip_object* objs = b.objs;
	const int64_t n = b.n_objs;
	const int leaf_size = std::max(1, std::min(7, opt.gpu_leaf));
	std::vector<int32_t> rest, big;
	for (int64_t i = 0; i < n; i++) if (b.can_be_hit((int32_t)i)) rest.push_back((int32_t)i);
	// the same few oversized primitives the surface-area build keeps out of its tree
	auto diag = [&](int32_t id) {
		Box bb = b.bounds64(id);
		double dx = bb.hi[0] - bb.lo[0], dy = bb.hi[1] - bb.lo[1], dz = bb.hi[2] - bb.lo[2];
		return std::sqrt(dx * dx + dy * dy + dz * dz);
	};
	if (rest.size() > (size_t)std::max(0, opt.big_leaf_min) && !opt.no_big_leaf) {
		std::vector<double> d(rest.size());
		for (size_t i = 0; i < rest.size(); i++) d[i] = diag(rest[i]);
		std::vector<double> all = d;
		std::nth_element(d.begin(), d.begin() + (long)(d.size() / 2), d.end());
		const double limit = 64.0 * d[d.size() / 2];
		int ns = 0, nc = 0;
		for (size_t i = 0; i < rest.size(); i++) if (all[i] > limit) { big.push_back(rest[i]); (objs[rest[i]].kind == RTHIP_SPHERE ? ns : nc)++; }
		if (ns > 7 || nc > 7) big.clear();
		if (!big.empty()) {
			size_t keep = 0;
			for (size_t i = 0; i < rest.size(); i++) if (!(all[i] > limit)) rest[keep++] = rest[i];
			rest.resize(keep);
		}
	}
	const size_t n_leaves = (rest.size() + (size_t)leaf_size - 1) / (size_t)leaf_size;
	if (n_leaves < 2) return false;
	// a leaf as emit_leaf makes it, or only listed
	auto place = [&](const std::vector<int32_t>& ids, int32_t first, bool as_leaf) {
		if (!slot_ids) return b.emit_leaf(ids.data(), (int)ids.size());
		Builder::Sub sub{0, empty_box(), empty_box()};
		int ns = 0, nc = 0;
		slot_ids->reserve(slot_ids->size() + ids.size());
		for (int pass = 0; pass < 2; pass++)
			for (int32_t id : ids) {
				if ((objs[id].kind == RTHIP_SPHERE) != (pass == 0)) continue;
				if (pass == 0) ns++; else nc++;
				slot_ids->push_back(id);
				if (!as_leaf) continue;  // only the leaf of oversized primitives is used as a leaf (code and boxes)
				sub.b64 = box_union(sub.b64, b.bounds64(id));
				sub.b32 = box_union(sub.b32, b.bounds32(id));
			}
		if (as_leaf) sub.code = leaf_code(first, ns, nc);
		return sub;
	};
	FlatScene::GpuPlan& g = out->gpu;
	g.on = true;
	g.leaf_size = leaf_size;
	g.n_leaves = (int32_t)n_leaves;
	g.node_base = big.empty() ? 0 : 1;
	const size_t n_nodes = (size_t)g.node_base + n_leaves - 1;
	g.n_nodes = (int32_t)n_nodes;
	// (listing only: the nodes are the device's to allocate; the host composes node 0 alone)
	out->nodes64.assign(slot_ids ? 1 : n_nodes, Node<double>{});
	out->nodes32.assign(slot_ids ? 1 : n_nodes, Node<float>{});
	if (!big.empty()) {
		Builder::Sub r;
		r.code = leaf_code(0, 0, 0);
		r.b64 = r.b32 = empty_box();
		Builder::Sub bl = place(big, 0, true);
		b.write_node(0, bl, r);  // the right child is the device's to fill in
		out->big_leaf = bl.code;
		out->start = 1;
	}
	g.first = (int32_t)big.size();
	g.count = (int32_t)rest.size();
	(void)place(rest, g.first, false);  // every other primitive, spheres then cubes, in object order
	double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
	for (int32_t id : rest)
		for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], objs[id].center[k]); hi[k] = std::fmax(hi[k], objs[id].center[k]); }
	// One scale for the three axes (the longest extent): the cells of the code are cubes, so a
	// flat scene is not cut into slabs along its short axis (per-axis scales traced at half the rate).
	const double extent = std::fmax(std::fmax(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
	for (int k = 0; k < 3; k++) { g.clo[k] = lo[k]; g.cinv[k] = extent > 0.0 ? 1.0 / extent : 0.0; }
	g.quantise = n_nodes > 2048 && !opt.no_qnodes;
	if (g.quantise && !slot_ids) out->qnodes.assign(n_nodes, QNode{});
	return true;
}
// Materials in primitive order: shading reads the record of the primitive that was hit without a
// trip through the primitive's object id, and small scenes keep the array in LDS beside the primitives.
void materials_in_primitive_order(FlatScene* out) {
	std::vector<Mat<float>> m32(out->prims32.size());
	std::vector<Mat<double>> m64(out->prims64.size());
	for (size_t i = 0; i < out->prims32.size(); i++) {
		const size_t id = (size_t)(out->prims32[i].id & ((1 << kMatShift) - 1));
		m32[i] = out->mats32[id];
		m64[i] = out->mats64[id];
	}
	out->mats32.swap(m32);
	out->mats64.swap(m64);
}
}  // namespace

bool build_flat_scene(const rthip_object* objs, int64_t n, FlatScene* out, std::string* err, const BuildOptions& opt) {
	Builder b;
	const bool timing = opt.timing;
	auto t_start = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if (!timing) return;
		auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "[rthip build] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_start).count());
		t_start = now;
	};
	if (!reference_pass(objs, n, out, err, opt, &b, lap)) return false;
	const int ref_root = b.ref_root;
	// 2. the device tree
	bool use_sah = !opt.tree_ref;
	b.max_leaf = std::max(1, std::min(7, opt.sah_leaf));
	b.kind_cap = std::max(0, std::min(7, opt.sah_kind_cap));
	if (opt.sah_cprim >= 0.0) b.cost_prim = opt.sah_cprim;
	if (opt.sah_sweep >= 0) b.sweep_max = opt.sah_sweep;
	bool big_ids_empty = true;
	int32_t root = 0;
	Builder::Sub l, r;
	r.code = leaf_code(0, 0, 0);
	r.b64 = r.b32 = empty_box();
	// 2a. tree_gpu: lay the primitives out and leave the tree to the device
	if (opt.tree_gpu && layout_for_device_build(b, opt, nullptr)) lap("layout for the device build");
	if (out->gpu.on) {
		// nothing more to build here
	} else if (use_sah) {
		auto attempt = [&](double cost_prim) {
			out->nodes64.clear(); out->nodes32.clear(); out->prims64.clear(); out->prims32.clear();
			b.items.clear();
			b.max_depth = 0;
			b.cost_prim = cost_prim;
			big_ids_empty = true;
			l = Builder::Sub();
			root = b.new_node();  // node 0 is the root
			for (int64_t i = 0; i < n; i++) {
				if (!b.can_be_hit((int32_t)i)) continue;
				Builder::Item it;
				it.id = (int32_t)i;
				it.b = b.bounds64((int32_t)i);
				for (int k = 0; k < 3; k++) it.c[k] = 0.5 * (it.b.lo[k] + it.b.hi[k]);
				b.items.push_back(it);
			}
			size_t m = b.items.size();
			// Small scenes afford the exact sweep everywhere (+3 % on a 500-object scene).  On a million
			// objects it costs a second of build time and a level of depth (the traversal stack is LDS
			// that resident waves pay for) without tracing faster.
			if (m <= 65536 && opt.sah_sweep < 0) b.sweep_max = 4096;
			// A few primitives that dwarf the rest (a ground slab, a sun) would stretch the box of every
			// node above them.  They go into one leaf of their own, the left child of the root, and the
			// surface-area build runs over the rest (its root is the right child).  Measured: +2.5 % on
			// the 500-object scenes, +4 % on the million-object one.
			std::vector<int32_t> big;
			if (m > (size_t)std::max(0, opt.big_leaf_min) && !opt.no_big_leaf) {
				auto diag = [](const Builder::Item& it) {
					double dx = it.b.hi[0] - it.b.lo[0], dy = it.b.hi[1] - it.b.lo[1], dz = it.b.hi[2] - it.b.lo[2];
					return std::sqrt(dx * dx + dy * dy + dz * dz);
				};
				std::vector<double> d(m);
				for (size_t i = 0; i < m; i++) d[i] = diag(b.items[i]);
				std::nth_element(d.begin(), d.begin() + (long)(m / 2), d.end());
				const double limit = 64.0 * d[m / 2];
				int ns = 0, nc = 0;
				for (size_t i = 0; i < m; i++)
					if (diag(b.items[i]) > limit) { big.push_back(b.items[i].id); (objs[b.items[i].id].kind == RTHIP_SPHERE ? ns : nc)++; }
				if (big.empty() || ns > 7 || nc > 7) {
					big.clear();  // none, or more than one leaf code can hold: leave the tree as it is
				} else {
					b.items.erase(std::remove_if(b.items.begin(), b.items.end(), [&](const Builder::Item& it) { return diag(it) > limit; }), b.items.end());
					m = b.items.size();
				}
			}
			big_ids_empty = big.empty();
			if (!big.empty()) {
				Builder::Sub bl = b.emit_leaf(big.data(), (int)big.size());  // node 0 is already reserved for the root
				Builder::Sub rest = b.sah_build(0, m, 1);
				b.write_node(root, bl, rest);
				if (b.max_depth < 1) b.max_depth = 1;
			} else if (m <= (size_t)b.max_leaf) {
				l = m ? b.sah_leaf(0, m) : r;
			} else {
				// the root's two children are the two halves of the first split
				out->nodes64.pop_back();
				out->nodes32.pop_back();
				Builder::Sub whole = b.sah_build(0, m, 0);
				(void)whole;
				root = 0;
			}
			if (big.empty() && m <= (size_t)b.max_leaf) { b.write_node(root, l, r); b.max_depth = 1; }
		};
		// Waves a CU keeps resident when every workgroup holds the f32 scene and its traversal stacks
		// in LDS (160 KB; 7 workgroups of 256 or 3 of 512 threads at most by registers).  Going from 24
		// to 16 resident waves costs 17 % on the default scene, far more than a coarser tree does.
		auto resident_waves = [&]() {
			auto up = [](size_t v) { return (v + 63) / 64 * 64; };
			size_t scene = up(out->nodes32.size() * sizeof(Node<float>)) + up(out->prims32.size() * sizeof(Prim<float>)) + up(out->gates32.size() * sizeof(Gate<float>));
			if (scene > 64 * 1024) return 0;
			// (Launches may pick two workgroups of 768 threads with the materials in LDS as well when that keeps
			// as many waves; the tree is sized for three of 512, which measured best: the room 768 threads
			// free is worth more as material storage than as a finer tree.)
			int best = 0;
			for (int threads : {256, 512}) {
				size_t per = (size_t)std::max(1, b.max_depth) * (size_t)threads * 4 + scene;
				int blocks = std::min(28 / (threads / 64), (int)(160 * 1024 / per));
				best = std::max(best, blocks * threads / 64);
			}
			return best;
		};
		size_t n_hittable = 0;
		for (int64_t i = 0; i < n; i++) n_hittable += b.can_be_hit((int32_t)i) ? 1 : 0;
		if (opt.sah_cprim >= 0.0 || n_hittable * sizeof(Prim<float>) > 64 * 1024) {
			// As asked; or a scene for global memory, where a primitive test is two trips to L1 in a
			// loop that runs a third full: leaves of one primitive nearly everywhere (cost 1.5) trace
			// 7 % faster on a million objects than the leaves of two that 0.4 gives.
			if (opt.sah_cprim < 0.0) b.cost_prim = 1.5;
			attempt(b.cost_prim);
		} else {
			// The finer the tree, the fewer primitive tests per ray (up to a cost of about one
			// node visit per test); take the finest one that keeps the CU full, else the fullest.
			static const double ladder[] = {0.7, 0.5, 0.4, 0.3, 0.2};
			double pick = ladder[0];
			int pick_waves = -1;
			for (double c : ladder) {
				attempt(c);
				int w = resident_waves();
				if (w > pick_waves) { pick_waves = w; pick = c; }
				if (w >= 24) break;
			}
			if (pick_waves == 0) pick = 1.5;  // no tree fits LDS: the value tuned for global memory (see above)
			if (b.cost_prim != pick) attempt(pick);
		}
	} else {
		root = b.new_node();
		if (b.ref_nodes[(size_t)ref_root].left < 0) {
			l = b.shape_ref(ref_root, 0);
			b.write_node(root, l, r);
			b.max_depth = 1;
		} else {
			out->nodes64.pop_back();
			out->nodes32.pop_back();
			b.shape_ref(ref_root, 0);
		}
	}
	lap("device tree");
	materials_in_primitive_order(out);
	if (out->gpu.on) return true;  // depth, start and the quantised nodes come from the device build
	out->depth = std::max(1, b.max_depth);
	if (use_sah && !big_ids_empty) { out->big_leaf = out->nodes32[0].child[0]; out->start = out->nodes32[0].child[1]; }
	if (out->depth > kStackMax) { *err = "tree deeper than the traversal stack"; return false; }
	// 3. large scenes: the quantised copy of the f32 nodes (see QNode).  The grid spans the boxes
	// of the surface-area tree; the leaf of oversized primitives stays outside it (node 0, which
	// holds that leaf and the tree's root, is then not part of the quantised traversal).
	if (use_sah && out->nodes32.size() > 2048 && !opt.no_qnodes) {
		const Node<float>& n0 = out->nodes32[0];
		const bool has_big = !big_ids_empty;
		// planes of child c of an f32 node (stored as centre and half extent) on axis k; an empty box has lo > hi
		auto plane_lo = [](const Node<float>& n, int c, int k) {
			static const int lo_slot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, hi_slot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
			return n.box[6 * c + hi_slot[k]] < 0.0f ? (double)INFINITY : (double)n.box[6 * c + lo_slot[k]] - (double)n.box[6 * c + hi_slot[k]];
		};
		auto plane_hi = [](const Node<float>& n, int c, int k) {
			static const int lo_slot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, hi_slot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
			return n.box[6 * c + hi_slot[k]] < 0.0f ? -(double)INFINITY : (double)n.box[6 * c + lo_slot[k]] + (double)n.box[6 * c + hi_slot[k]];
		};
		double lo[3], hi[3];
		for (int k = 0; k < 3; k++) {
			lo[k] = has_big ? plane_lo(n0, 1, k) : std::fmin(plane_lo(n0, 0, k), plane_lo(n0, 1, k));
			hi[k] = has_big ? plane_hi(n0, 1, k) : std::fmax(plane_hi(n0, 0, k), plane_hi(n0, 1, k));
		}
		QFrame& f = out->qframe;
		for (int k = 0; k < 3; k++) {
			f.centre[k] = (float)(0.5 * (lo[k] + hi[k]));
			double half = std::fmax(hi[k] - (double)f.centre[k], (double)f.centre[k] - lo[k]);
			f.step[k] = (float)std::fmax(half / (kQMax - 64), 1e-30);  // headroom for the outward rounding
		}
		f.start = has_big ? n0.child[1] : 0;
		f.big_leaf = has_big ? n0.child[0] : 0;
		out->qnodes.resize(out->nodes32.size());
		bool ok = true;
		for (size_t i = has_big ? 1 : 0; i < out->nodes32.size(); i++) {
			const Node<float>& n = out->nodes32[i];
			QNode& q = out->qnodes[i];
			for (int c = 0; c < 2; c++) {
				for (int k = 0; k < 3; k++) {
					static const int lo_slot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, hi_slot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
					double l = plane_lo(n, c, k), h = plane_hi(n, c, k);
					if (!(l <= h)) { q.q[6 * c + lo_slot[k]] = kQMax; q.q[6 * c + hi_slot[k]] = 0; continue; }  // empty box: a far corner point (it leads to an empty leaf anyway)
					double ql = std::floor((l - (double)f.centre[k]) / (double)f.step[k]) - 1.0;
					double qh = std::ceil((h - (double)f.centre[k]) / (double)f.step[k]) + 1.0;
					if (ql < -kQMax || qh > kQMax) ok = false;
					ql = std::fmax(ql, -kQMax); qh = std::fmin(qh, kQMax);
					q.q[6 * c + lo_slot[k]] = (int16_t)ql;
					q.q[6 * c + hi_slot[k]] = (int16_t)(uint16_t)(qh - ql);  // width, unsigned
				}
				q.child[c] = n.child[c];
			}
		}
		if (!ok) out->qnodes.clear();  // cannot happen with the headroom above; the full nodes are always valid
	}
	return true;
}

bool refit_flat_scene(const rthip_object* objs, int64_t n, const int32_t* slot_ids, int64_t n_slots, FlatScene* out, std::string* err, const BuildOptions& opt) {
	Builder b;
	if (!reference_pass(objs, n, out, err, opt, &b, [](const char*) {})) return false;
	int64_t hittable = 0;
	for (int64_t i = 0; i < n; i++) hittable += b.can_be_hit((int32_t)i) ? 1 : 0;
	if (hittable != n_slots) { *err = "refit: " + std::to_string(hittable) + " objects can be hit, the tree holds " + std::to_string(n_slots) + " (upload the scene instead)"; return false; }
	std::vector<bool> seen((size_t)n, false);
	for (int64_t s = 0; s < n_slots; s++) {
		const int32_t old = slot_ids[s], id = old & ((1 << kMatShift) - 1);
		if (id < 0 || id >= n || seen[(size_t)id] || !b.can_be_hit(id) || ((old >> kKindShift) & 1) != objs[id].kind) {
			*err = "refit: object " + std::to_string(id) + " is not the kind of primitive the tree holds in its place, or can no longer be hit (upload the scene instead)";
			return false;
		}
		seen[(size_t)id] = true;
		b.emit_prim(id);
	}
	materials_in_primitive_order(out);
	return true;
}

bool plan_device_scene(const rthip_object* objs, int64_t n, FlatScene* out, std::vector<int32_t>* slot_ids, std::string* err, const BuildOptions& opt) {
	*out = FlatScene();
	slot_ids->clear();
	if (!check_objects(objs, n, err)) return false;
	Builder b;
	b.objs = objs;
	b.n_objs = n;
	b.out = out;
	(void)layout_for_device_build(b, opt, slot_ids);
	return true;
}

bool check_scene_objects(const rthip_object* objs, int64_t n, std::string* err, int64_t* hittable) { return check_objects(objs, n, err, hittable); }

uint64_t checksum_flat_scene(const FlatScene& s) {
	uint64_t h = 1469598103934665603ull;
	auto mix = [&](const void* p, size_t n) {
		const unsigned char* b = static_cast<const unsigned char*>(p);
		for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
	};
	// field by field: the structs have padding bytes that hold nothing
	for (const auto& n : s.nodes32) { mix(n.box, sizeof(n.box)); mix(n.child, sizeof(n.child)); }
	for (const auto& n : s.nodes64) { mix(n.box, sizeof(n.box)); mix(n.child, sizeof(n.child)); }
	for (const auto& p : s.prims32) { mix(p.v, sizeof(p.v)); mix(&p.id, 4); mix(&p.rank, 4); }
	for (const auto& p : s.prims64) { mix(p.v, sizeof(p.v)); mix(&p.id, 4); mix(&p.rank, 4); }
	for (const auto& q : s.qnodes) { mix(q.q, sizeof(q.q)); mix(q.child, sizeof(q.child)); }
	for (const auto& g : s.gates64) mix(g.box, sizeof(g.box));
	mix(s.leaf_ids.data(), s.leaf_ids.size() * 4);
	mix(&s.depth, sizeof(s.depth));
	return h;
}

int64_t validate_flat_scene(const FlatScene& s, int64_t n_objects, std::string* first) {
	int64_t bad = 0;
	auto report = [&](const std::string& what) { if (bad++ == 0 && first) *first = what; };
	struct B { double lo[3], hi[3]; };
	auto prim_bounds = [&](const Prim<float>& p) {
		B b;
		const bool sphere = ((p.id >> kKindShift) & 1) == RTHIP_SPHERE;
		for (int k = 0; k < 3; k++) {
			static const int lo_slot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, hi_slot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
			if (sphere) { b.lo[k] = (double)p.v[k] - std::fabs((double)p.v[3]); b.hi[k] = (double)p.v[k] + std::fabs((double)p.v[3]); }
			else { b.lo[k] = p.v[lo_slot[k]]; b.hi[k] = p.v[hi_slot[k]]; }
		}
		return b;
	};
	std::vector<int> seen((size_t)n_objects, 0);
	std::vector<char> prim_used(s.prims32.size(), 0);
	const bool quantised = !s.qnodes.empty();
	int deepest = 0;
	// bounds of the subtree behind a child code; checks every box on the way up
	std::function<B(int32_t, int)> walk = [&](int32_t code, int depth) -> B {
		B b;
		for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
		if (code < 0) {
			const uint32_t c = ~(uint32_t)code;
			const int32_t first_prim = (int32_t)(c >> 6), cnt = (int32_t)((c >> 3) & 7u) + (int32_t)(c & 7u);
			for (int32_t i = first_prim; i < first_prim + cnt; i++) {
				if (i < 0 || (size_t)i >= s.prims32.size()) { report("leaf points outside the primitive array"); continue; }
				if (prim_used[(size_t)i]++) report("primitive " + std::to_string(i) + " is in two leaves");
				const int32_t id = s.prims32[(size_t)i].id & ((1 << kMatShift) - 1);
				if (id < 0 || id >= n_objects) report("primitive with an object id out of range"); else seen[(size_t)id]++;
				B pb = prim_bounds(s.prims32[(size_t)i]);
				for (int k = 0; k < 3; k++) { b.lo[k] = std::fmin(b.lo[k], pb.lo[k]); b.hi[k] = std::fmax(b.hi[k], pb.hi[k]); }
			}
			return b;
		}
		if ((size_t)code >= s.nodes32.size()) { report("child index outside the node array"); return b; }
		deepest = std::max(deepest, depth + 1);
		const Node<float>& n = s.nodes32[(size_t)code];
		for (int c = 0; c < 2; c++) {
			B cb = walk(n.child[c], depth + 1);
			for (int k = 0; k < 3; k++) {
				static const int lo_slot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, hi_slot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
				if (!(cb.lo[k] <= cb.hi[k])) continue;  // nothing below: any box will do
				const double centre = n.box[6 * c + lo_slot[k]], half = n.box[6 * c + hi_slot[k]];
				if (!(centre - half <= cb.lo[k] && centre + half >= cb.hi[k]))
					report("node " + std::to_string(code) + " child " + std::to_string(c) + " axis " + std::to_string(k) + ": f32 box does not contain its primitives");
				if (quantised && !(code == 0 && s.big_leaf != 0)) {
					const QNode& q = s.qnodes[(size_t)code];
					const double lo = (double)s.qframe.centre[k] + (double)q.q[6 * c + lo_slot[k]] * (double)s.qframe.step[k];
					const double hi = lo + (double)(uint16_t)q.q[6 * c + hi_slot[k]] * (double)s.qframe.step[k];
					if (!(lo <= cb.lo[k] && hi >= cb.hi[k]))
						report("node " + std::to_string(code) + " child " + std::to_string(c) + " axis " + std::to_string(k) + ": quantised box does not contain its primitives");
					if (q.child[c] != n.child[c]) report("quantised node " + std::to_string(code) + " has other children than the full node");
				}
				b.lo[k] = std::fmin(b.lo[k], cb.lo[k]); b.hi[k] = std::fmax(b.hi[k], cb.hi[k]);
			}
		}
		return b;
	};
	if (s.nodes32.empty()) { report("no nodes"); return bad; }
	walk(0, 0);
	if (deepest > s.depth) report("tree deeper (" + std::to_string(deepest) + ") than reported (" + std::to_string(s.depth) + ")");
	for (size_t i = 0; i < prim_used.size(); i++) if (!prim_used[i]) report("primitive " + std::to_string(i) + " is in no leaf");
	for (int64_t i = 0; i < n_objects; i++) if (seen[(size_t)i] > 1) report("object " + std::to_string(i) + " appears more than once");
	if (s.big_leaf != 0 && (s.nodes32[0].child[0] != s.big_leaf || s.nodes32[0].child[1] != s.start)) report("big_leaf/start do not match node 0");
	if (s.nodes32.size() != s.nodes64.size() || s.prims32.size() != s.prims64.size()) report("f32 and f64 arrays differ in size");
	for (size_t i = 0; i < s.nodes32.size() && i < s.nodes64.size(); i++)
		if (s.nodes32[i].child[0] != s.nodes64[i].child[0] || s.nodes32[i].child[1] != s.nodes64[i].child[1]) { report("f32 and f64 trees differ in shape"); break; }
	return bad;
}

}  // namespace rthip
