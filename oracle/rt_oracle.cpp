// rt_oracle.cpp — TEST INFRASTRUCTURE ONLY (never linked or loaded by the product path).
//
// CPU restatement, in f64, of the per-pixel path-tracing hot path of Peanutt42/raytracer.
// Every function cites the reference file:line it follows and keeps the reference's
// operation order, so that results agree with the Rust code to the last bit wherever the
// Rust code itself is deterministic.  Compile with -ffp-contract=off (see Makefile).
//
// Parity status: PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures for
// this path and cannot be compiled in this environment (no Rust toolchain), so nothing produced
// by the reference itself anchors this restatement: it rests on source reading plus closed-form
// known-answer tests.  The random stream is unpinned by design: the reference draws from an
// OS-seeded ThreadRng
// (raytracer_headless/src/main.rs:40), so only the distributions and the per-event draw
// counts are contractual; the stream used here is the counter-based generator that the HIP
// kernels also implement.
//
// Two traversal modes are provided:
//   FAITHFUL: the reference's algorithm literally (src/bvh.rs:84-127 recursive, both
//             children visited, the per-axis AABB test of src/math.rs:356-371, recursive
//             ray_color of src/renderer.rs:3-26, powf in the Schlick term).
//   TIGHT:    same per-primitive arithmetic and the same tie rules, but a true slab test,
//             near-child-first ordering with entry_t > best_t culling, and the iterative
//             form of ray_color.  This is the algorithm the GPU kernels implement; tests
//             assert FAITHFUL == TIGHT on every scene.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

typedef double Scalar;  // src/math.rs:4

// Rust's f64::min / f64::max (src/math.rs:103, src/shapes/cube.rs:34-35) return the other operand when
// one is NaN, which is what C's fmin / fmax do; written out here because without -ffinite-math-only the
// compiler calls libm for them (five nanoseconds a call, a dozen calls per box test).  Which zero comes
// back for (+0, -0) is not defined by either and decides nothing here: results are only compared.
static inline Scalar fmin_(Scalar a, Scalar b) { return a < b ? a : (b != b ? a : b); }
static inline Scalar fmax_(Scalar a, Scalar b) { return a > b ? a : (b != b ? a : b); }

// ---------------------------------------------------------------------------------------
// src/math.rs:18-318  Vec3 and its operators
// ---------------------------------------------------------------------------------------
struct Vec3 {
	Scalar x, y, z;
	Scalar operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }  // math.rs:294-306
};
static inline Vec3 V(Scalar x, Scalar y, Scalar z) { return Vec3{x, y, z}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }  // :177-188
static inline Vec3 operator-(Vec3 a, Vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }  // :190-201
static inline Vec3 operator*(Vec3 a, Scalar s) { return V(a.x * s, a.y * s, a.z * s); }      // :203-214
static inline Vec3 operator*(Scalar s, Vec3 v) { return V(v.x * s, v.y * s, v.z * s); }      // :216-227
static inline Vec3 operator*(Vec3 a, Vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }  // :229-240
static inline Vec3 operator/(Vec3 a, Scalar s) { return V(a.x / s, a.y / s, a.z / s); }      // :255-266
static inline Vec3 operator/(Scalar s, Vec3 v) { return V(s / v.x, s / v.y, s / v.z); }      // :268-279
static inline Vec3 operator-(Vec3 a) { return V(-a.x, -a.y, -a.z); }                         // :281-292
static inline Scalar dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // :54-56
static inline Vec3 cross(Vec3 a, Vec3 b) {                                                   // :59-65
	return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline Scalar length_squared(Vec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }  // :68-70
static inline Scalar length(Vec3 a) { return std::sqrt(length_squared(a)); }               // :73-75
static inline Vec3 normalize(Vec3 a) {                                                     // :78-85
	Scalar l = length(a);
	return V(a.x / l, a.y / l, a.z / l);
}
static inline Vec3 vabs(Vec3 a) { return V(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }  // :88-94
// math.rs:97-99  `*self - (normal * 2.0 * self.dot(normal))`, i.e. ((normal*2.0)*dot)
static inline Vec3 reflect(Vec3 v, Vec3 n) { return v - ((n * 2.0) * dot(v, n)); }
// math.rs:102-107 ; Rust f64::min ignores a NaN operand, as fmin does.
static inline Vec3 refract(Vec3 uv, Vec3 n, Scalar etai_over_etat) {
	Scalar cos_theta = fmin_(dot(-uv, n), 1.0);
	Vec3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
	Vec3 r_out_parallel = -(std::sqrt(std::fabs(1.0 - length_squared(r_out_perp)))) * n;
	return r_out_perp + r_out_parallel;
}
// math.rs:110-117
static inline bool near_zero(Vec3 a) {
	const Scalar tol = 0.00000001;
	return (std::fabs(a.x) <= tol) && (std::fabs(a.y) <= tol) && (std::fabs(a.z) <= tol);
}
// math.rs:147-153
static inline Vec3 linear_to_gamma(Vec3 a) { return V(std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)); }
// math.rs:8-10
static inline Scalar radians(Scalar degrees) { return degrees * M_PI / 180.0; }

struct Ray {  // math.rs:321-336
	Vec3 origin, dir;
	Vec3 at(Scalar t) const { return origin + dir * t; }
};

struct AABB {  // math.rs:339-403
	Vec3 min, max;
};
// math.rs:356-371.  min/max are per-axis locals: every axis is clipped against the constants
// [0.001, f64::MAX] independently; the interval is not carried from one axis to the next.
static inline bool aabb_hit_reference(const AABB& b, const Ray& ray) {
	for (int a = 0; a < 3; a++) {
		Scalar inv_d = 1.0 / ray.dir[a];
		Scalar t0 = (b.min[a] - ray.origin[a]) * inv_d;
		Scalar t1 = (b.max[a] - ray.origin[a]) * inv_d;
		if (inv_d < 0.0) std::swap(t0, t1);
		Scalar mn = t0 > 0.001 ? t0 : 0.001;
		Scalar mx = t1 < DBL_MAX ? t1 : DBL_MAX;
		if (mx <= mn) return false;
	}
	return true;
}
// math.rs:389-402 ; Rust f64::min/max ignore NaN like fmin/fmax.
static inline AABB surrounding(const AABB& a, const AABB& b) {
	return AABB{V(fmin_(a.min.x, b.min.x), fmin_(a.min.y, b.min.y), fmin_(a.min.z, b.min.z)),
	            V(fmax_(a.max.x, b.max.x), fmax_(a.max.y, b.max.y), fmax_(a.max.z, b.max.z))};
}

// ---------------------------------------------------------------------------------------
// Random stream.  The reference uses rand 0.10 ThreadRng (src/math.rs:13-15,156-162); its
// output is OS-seeded and therefore not a contract.  This generator is the specification
// shared (by description, not by code) with the HIP kernels: a PCG RXS-M-XS 32-bit stream
// whose state and increment are hashed from (seed, pixel index, sample index).
// One draw = one 32-bit word r; the f64 uniform is r * 2^-32 in [0,1).
// ---------------------------------------------------------------------------------------
static inline uint32_t mix32(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
struct PathRng {
	uint32_t state, inc;
	uint64_t draws;
	void init(uint32_t seed, uint32_t pixel, uint32_t sample) {
		uint32_t a = mix32(seed + 0x9E3779B9u * (pixel + 1u));
		uint32_t b = mix32(a ^ (0x85EBCA6Bu * (sample + 1u)));
		state = b;
		inc = (mix32(b + a + 0xC2B2AE35u) << 1) | 1u;
		draws = 0;
	}
	uint32_t next_u32() {
		uint32_t old = state;
		state = old * 747796405u + inc;
		uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
		draws++;
		return (word >> 22u) ^ word;
	}
	Scalar uniform01() { return (Scalar)next_u32() * (1.0 / 4294967296.0); }
};
// math.rs:13-15  random(min,max) = random_range(min..max): half-open, uniform*scale + low.
static inline Scalar random_range(Scalar mn, Scalar mx, PathRng& r) { return mn + (mx - mn) * r.uniform01(); }
// math.rs:156-162  three draws in x, y, z order.
static inline Vec3 vec3_random(Scalar mn, Scalar mx, PathRng& r) {
	Scalar x = mn + (mx - mn) * r.uniform01();
	Scalar y = mn + (mx - mn) * r.uniform01();
	Scalar z = mn + (mx - mn) * r.uniform01();
	return V(x, y, z);
}
// math.rs:165-168  a cube sample pushed to the sphere (not uniform on the sphere: contractual).
static inline Vec3 random_unit_vector(PathRng& r) { return normalize(vec3_random(-1.0, 1.0, r)); }
// math.rs:171-174  a square sample pushed to the unit circle (not the disk: contractual).
static inline Vec3 random_in_unit_disk(PathRng& r) {
	Scalar x = random_range(-1.0, 1.0, r);
	Scalar y = random_range(-1.0, 1.0, r);
	return normalize(V(x, y, 0.0));
}

// ---------------------------------------------------------------------------------------
// src/materials.rs
// ---------------------------------------------------------------------------------------
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2 };
struct Material {  // materials.rs:4-8
	int type;
	Vec3 albedo;
	Scalar param;  // Lambertian: emission, Metal: fuzz, Dielectric: ir
};
struct RayHit {  // scene.rs:18-34
	Vec3 point, normal;
	const Material* material;
	bool front_face;
	int object_id;
	Scalar t;
};
// materials.rs:80-85.  use_pow selects powf(5.) (reference) or a multiply chain (GPU form).
static inline Scalar reflectance(Scalar cosine, Scalar ref_idx, bool use_pow) {
	Scalar r0 = (1. - ref_idx) / (1. + ref_idx);
	r0 = r0 * r0;
	Scalar x = 1. - cosine;
	Scalar p5;
	if (use_pow) {
		p5 = std::pow(x, 5.);
	} else {
		Scalar x2 = x * x;
		p5 = (x2 * x2) * x;
	}
	return r0 + (1. - r0) * p5;
}
struct Scattered {  // materials.rs:10-13
	Vec3 attenuation;
	Ray scattered;
};
// materials.rs:17-68.  Returns false for the reference's None.
static inline bool scatter(const Material& m, const Ray& ray_in, const RayHit& hit, PathRng& rng, bool use_pow,
                           Scattered* out) {
	switch (m.type) {
	case MAT_LAMBERTIAN: {  // :24-33
		Vec3 scatter_direction = hit.normal + random_unit_vector(rng);
		if (near_zero(scatter_direction)) scatter_direction = hit.normal;
		out->attenuation = m.albedo;
		out->scattered = Ray{hit.point, scatter_direction};
		return true;
	}
	case MAT_METAL: {  // :34-45
		Vec3 reflected = reflect(normalize(ray_in.dir), hit.normal);
		Vec3 scattered_dir = reflected + m.param * random_unit_vector(rng);
		if (dot(scattered_dir, hit.normal) <= 0.0) return false;
		out->attenuation = m.albedo;
		out->scattered = Ray{hit.point, scattered_dir};
		return true;
	}
	default: {  // Dielectric :46-66
		Scalar ir = m.param;
		Scalar refraction_ratio = hit.front_face ? 1.0 / ir : ir;
		Vec3 unit_dir = normalize(ray_in.dir);
		Scalar cos_theta = fmin_(dot(-unit_dir, hit.normal), 1.0);
		Scalar sin_theta = std::sqrt(1.0 - (cos_theta * cos_theta));
		bool cannot_refract = refraction_ratio * sin_theta > 1.0;
		// `||` short-circuits: the draw happens only when the ray can refract (:54-55).
		Vec3 direction;
		if (cannot_refract || (reflectance(cos_theta, refraction_ratio, use_pow) > random_range(0.0, 1.0, rng))) {
			direction = reflect(unit_dir, hit.normal);
		} else {
			direction = refract(unit_dir, hit.normal, refraction_ratio);
		}
		out->attenuation = V(1.0, 1.0, 1.0);
		out->scattered = Ray{hit.point, direction};
		return true;
	}
	}
}
// materials.rs:71-77
static inline Vec3 emission_color(const Material& m) {
	if (m.type == MAT_LAMBERTIAN) return m.albedo * m.param;
	return V(0.0, 0.0, 0.0);
}

// ---------------------------------------------------------------------------------------
// src/shapes/sphere.rs, src/shapes/cube.rs, src/scene.rs:36-113
// ---------------------------------------------------------------------------------------
enum { KIND_SPHERE = 0, KIND_CUBE = 1 };
struct Object {
	int kind;
	int id;  // insertion index in the scene = object id reported by the parity probe
	Vec3 center;
	Vec3 ext;  // sphere: ext.x = radius ; cube: half_extend
	Material material;
	AABB aabb;  // as the reference caches it (inverted when radius / half extent is negative)
	int rank;   // tie rank: larger wins at equal t (see assign_ranks)
};
static inline AABB object_aabb(int kind, Vec3 c, Vec3 ext) {
	if (kind == KIND_SPHERE) {  // sphere.rs:12-21
		Vec3 rv = V(ext.x, ext.x, ext.x);
		return AABB{c - rv, c + rv};
	}
	return AABB{c - ext, c + ext};  // cube.rs:12-19
}
// sphere.rs:26-48 : near root only; returns false for None.
static inline bool sphere_hit(const Object& s, const Ray& ray, Scalar* t) {
	Vec3 origin = ray.origin - s.center;
	Scalar a = dot(ray.dir, ray.dir);
	Scalar b = 2.0 * dot(origin, ray.dir);
	Scalar c = dot(origin, origin) - s.ext.x * s.ext.x;
	Scalar discriminant = b * b - 4.0 * a * c;
	if (discriminant < 0.0) return false;
	*t = (-b - std::sqrt(discriminant)) / (2.0 * a);
	return true;
}
// cube.rs:24-42 ; Rust f64::max/min ignore NaN like fmax/fmin.
static inline bool cube_hit(const Object& cb, const Ray& ray, Scalar* t) {
	Vec3 origin = ray.origin - cb.center;
	Vec3 m = 1.0 / ray.dir;
	Vec3 n = m * origin;
	Vec3 k = vabs(m) * cb.ext;
	Vec3 t1 = -n - k;
	Vec3 t2 = -n + k;
	Scalar t_n = fmax_(fmax_(t1.x, t1.y), t1.z);
	Scalar t_f = fmin_(fmin_(t2.x, t2.y), t2.z);
	if (t_n > t_f || t_f <= 0. || t_n <= 0.) return false;
	*t = t_n;
	return true;
}
static inline bool object_hit(const Object& o, const Ray& ray, Scalar* t) {  // scene.rs:42-50
	return o.kind == KIND_SPHERE ? sphere_hit(o, ray, t) : cube_hit(o, ray, t);
}
// Rust f64::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives, NaN for NaN.
static inline Scalar rust_signum(Scalar x) { return std::isnan(x) ? x : std::copysign(1.0, x); }
// sphere.rs:60-62 and cube.rs:54-64
static inline Vec3 object_normal(const Object& o, Vec3 p) {
	if (o.kind == KIND_SPHERE) return (p - o.center) / o.ext.x;
	Vec3 rel_p = p - o.center;
	Scalar maxc = fmax_(fmax_(std::fabs(rel_p.x), std::fabs(rel_p.y)), std::fabs(rel_p.z));
	if (maxc == std::fabs(rel_p.x)) return V(rust_signum(rel_p.x), 0.0, 0.0);
	if (maxc == std::fabs(rel_p.y)) return V(0.0, rust_signum(rel_p.y), 0.0);
	return V(0.0, 0.0, rust_signum(rel_p.z));
}

struct Counters {
	uint64_t rays, box_tests, sphere_tests, cube_tests, paths;
	uint64_t scatter_lambertian, scatter_metal, scatter_dielectric, sky_misses, rng_draws;
};
static inline void counters_add(Counters& a, const Counters& b) {
	a.rays += b.rays; a.box_tests += b.box_tests; a.sphere_tests += b.sphere_tests; a.cube_tests += b.cube_tests;
	a.paths += b.paths; a.scatter_lambertian += b.scatter_lambertian; a.scatter_metal += b.scatter_metal;
	a.scatter_dielectric += b.scatter_dielectric; a.sky_misses += b.sky_misses; a.rng_draws += b.rng_draws;
}

// scene.rs:100-113 : leaf linear scan, strict `<` so the first object wins a tie.
static inline const Object* scene_hit(const std::vector<Object>& objects, const Ray& ray, Scalar* t_out, Counters& c) {
	Scalar closest = DBL_MAX;
	const Object* closest_object = nullptr;
	for (const Object& o : objects) {
		if (o.kind == KIND_SPHERE) c.sphere_tests++; else c.cube_tests++;
		Scalar t;
		if (object_hit(o, ray, &t) && t > 0.001 && t < closest) {
			closest = t;
			closest_object = &o;
		}
	}
	*t_out = closest;
	return closest_object;
}
// scene.rs:116-120
static inline Vec3 sky_color_reference(Vec3 ray_dir) {
	Vec3 unit_dir = normalize(ray_dir);
	Scalar a = 0.5 * (unit_dir.y + 1.0);
	return V(1.0, 1.0, 1.0) * (1.0 - a) + V(0.5, 0.7, 1.0) * a;
}
// Wallpaper sky: the formula of raytracer_gpu/src/shaders/compute.wgsl:186-195 evaluated in
// f64 on the normalised direction.  It exists nowhere on the reference's CPU path; it is a
// render parameter of this build (sky_mode 1) so the wallpaper scene can be lit as intended.
static inline Vec3 sky_color_wallpaper(Vec3 ray_dir) {
	Vec3 d = normalize(ray_dir);
	Scalar strength = 0.5 * (-d.y + 0.25);
	Scalar t = 0.5 * (d.x + 1.0);
	Scalar e = (2.0 * t - 1.0);
	strength *= 50.0 * std::pow(99.0, e * e - 1.0);
	Vec3 a = V(0.94, 0.02, 0.99), b = V(0.0, 0.85, 0.98), c = V(0.0, 0.45, 0.98), dd = V(0.0, 0.98, 0.45);
	return (a * (1.0 - t) + b * t) * strength + (0.25 * c) * (0.5 * (d.y + 1.0)) + (0.15 * dd) * (0.5 * (-d.x + 1.0));
}
static inline Vec3 sky_color(Vec3 dir, int sky_mode) {
	return sky_mode == 1 ? sky_color_wallpaper(dir) : sky_color_reference(dir);
}

// ---------------------------------------------------------------------------------------
// src/bvh.rs
// ---------------------------------------------------------------------------------------
struct BVH {
	bool is_leaf;
	std::unique_ptr<BVH> left, right;  // Branch   (bvh.rs:6)
	std::vector<Object> objects;       // FewObjects(Scene) (bvh.rs:7)
	AABB aabb;                         // as the reference computes it (bvh.rs:13)
	AABB tight;                        // true bounds (|radius|, |half extent|) for TIGHT mode
};
static inline AABB true_bounds(const Object& o) {
	Vec3 e = o.kind == KIND_SPHERE ? V(std::fabs(o.ext.x), std::fabs(o.ext.x), std::fabs(o.ext.x)) : vabs(o.ext);
	return AABB{o.center - e, o.center + e};
}
// bvh.rs:17-81.  sort_unstable_by leaves the order of equal keys unspecified; this build
// fixes it as a stable sort of the current sequence (and the first axis of maximal range).
// tie_reverse = the opposite extreme: equal keys end up in REVERSED order (a stable sort of the
// reversed sequence), to measure how much the unspecified order can matter.
static std::unique_ptr<BVH> bvh_new(std::vector<Object> objects, bool tie_reverse = false) {
	if (objects.empty()) return nullptr;  // :51
	Scalar range[3];
	for (int axis = 0; axis < 3; axis++) {  // :28-43
		Scalar bmin = DBL_MAX, bmax = -DBL_MAX;
		for (const Object& o : objects) {
			bmin = fmin_(bmin, o.aabb.min[axis]);
			bmax = fmax_(bmax, o.aabb.max[axis]);
		}
		range[axis] = bmax - bmin;
	}
	int axis = 0;  // :45-47, descending by range; first maximal axis on ties
	for (int a = 1; a < 3; a++)
		if (range[a] > range[axis]) axis = a;
	if (tie_reverse) std::reverse(objects.begin(), objects.end());
	std::stable_sort(objects.begin(), objects.end(), [axis](const Object& a, const Object& b) {  // :18-26,49
		Scalar ac = a.aabb.min[axis] + a.aabb.max[axis];
		Scalar bc = b.aabb.min[axis] + b.aabb.max[axis];
		return ac < bc;
	});
	std::unique_ptr<BVH> node(new BVH());
	size_t len = objects.size();
	if (len < 6) {  // :52-61
		AABB aabb = objects[0].aabb;
		AABB tight = true_bounds(objects[0]);
		for (size_t i = 1; i < len; i++) {
			aabb = surrounding(aabb, objects[i].aabb);
			tight = surrounding(tight, true_bounds(objects[i]));
		}
		node->is_leaf = true;
		node->objects = std::move(objects);
		node->aabb = aabb;
		node->tight = tight;
		return node;
	}
	// :62-79  the upper half goes right, the rest left.
	std::vector<Object> upper(objects.begin() + len / 2, objects.end());
	objects.resize(len / 2);
	node->right = bvh_new(std::move(upper), tie_reverse);
	node->left = bvh_new(std::move(objects), tie_reverse);
	node->is_leaf = false;
	node->aabb = surrounding(node->left->aabb, node->right->aabb);
	node->tight = surrounding(node->left->tight, node->right->tight);
	return node;
}
// Tie ranks.  In the reference a leaf keeps the FIRST object among equal t (scene.rs:105,
// strict <) and a branch keeps the RIGHT child on equal t (bvh.rs:98-104).  Numbering leaves
// left to right and objects inside a leaf in reverse makes "largest rank among equal t" the
// reference's winner, whatever order a traversal visits them in.
static void assign_ranks(BVH* n, int* next) {
	if (n->is_leaf) {
		int k = (int)n->objects.size();
		for (int j = 0; j < k; j++) n->objects[j].rank = *next + (k - 1 - j);
		*next += k;
		return;
	}
	assign_ranks(n->left.get(), next);
	assign_ranks(n->right.get(), next);
}
static int bvh_depth(const BVH* n) { return n->is_leaf ? 1 : 1 + std::max(bvh_depth(n->left.get()), bvh_depth(n->right.get())); }
static void bvh_leaf_order(const BVH* n, std::vector<int>* ids, std::vector<int>* leaf_sizes) {
	if (n->is_leaf) {
		for (const Object& o : n->objects) ids->push_back(o.id);
		leaf_sizes->push_back((int)n->objects.size());
		return;
	}
	bvh_leaf_order(n->left.get(), ids, leaf_sizes);
	bvh_leaf_order(n->right.get(), ids, leaf_sizes);
}

// bvh.rs:84-127, literally.
static const Object* bvh_hit_faithful(const BVH* n, const Ray& ray, Scalar* t_out, Counters& c) {
	c.box_tests++;
	if (!aabb_hit_reference(n->aabb, ray)) return nullptr;
	auto is_distance_valid = [](Scalar d) { return d > 0.001; };
	if (n->is_leaf) return scene_hit(n->objects, ray, t_out, c);
	Scalar lt = 0, rt = 0;
	const Object* lo = bvh_hit_faithful(n->left.get(), ray, &lt, c);
	const Object* ro = bvh_hit_faithful(n->right.get(), ray, &rt, c);
	if (lo && ro) {
		if (is_distance_valid(lt) && lt < rt) { *t_out = lt; return lo; }
		if (is_distance_valid(rt)) { *t_out = rt; return ro; }
		return nullptr;
	}
	if (lo) { if (is_distance_valid(lt)) { *t_out = lt; return lo; } return nullptr; }
	if (ro) { if (is_distance_valid(rt)) { *t_out = rt; return ro; } return nullptr; }
	return nullptr;
}

// TIGHT mode: true slab test against (0.001, best_t], near child first, a subtree is skipped
// when its entry distance exceeds the best hit so far.  Candidates compare by (t, rank).
// The box distances and a primitive's own t are rounded differently (a cube's top face and
// the face of the box around it are the same plane, computed by two formulas), so the far
// side and the best_t bound are widened by kSlabPad: culling stays conservative and two
// coplanar faces still meet in the (t, rank) comparison.  2^-30 covers the cancellation of
// the reference's cube formula for slabs of half extent 1e5.
static const Scalar kSlabPad = 1.0 + 0x1p-30;
struct TightBest {
	Scalar t;
	const Object* obj;
};
static inline bool slab_test(const AABB& b, const Ray& ray, Vec3 inv, Scalar best_t, Scalar* entry) {
	Scalar tx0 = (b.min.x - ray.origin.x) * inv.x, tx1 = (b.max.x - ray.origin.x) * inv.x;
	Scalar ty0 = (b.min.y - ray.origin.y) * inv.y, ty1 = (b.max.y - ray.origin.y) * inv.y;
	Scalar tz0 = (b.min.z - ray.origin.z) * inv.z, tz1 = (b.max.z - ray.origin.z) * inv.z;
	Scalar tnear = fmax_(fmax_(fmin_(tx0, tx1), fmin_(ty0, ty1)), fmin_(tz0, tz1));
	Scalar tfar = fmin_(fmin_(fmax_(tx0, tx1), fmax_(ty0, ty1)), fmax_(tz0, tz1));
	*entry = tnear;
	tfar = tfar * kSlabPad;
	return tnear <= tfar && tfar > 0.001 && tnear <= best_t * kSlabPad;
}
static inline void tight_leaf(const BVH* n, const Ray& ray, TightBest& best, Counters& c) {
	for (const Object& o : n->objects) {
		if (o.kind == KIND_SPHERE) c.sphere_tests++; else c.cube_tests++;
		Scalar t;
		if (object_hit(o, ray, &t) && t > 0.001) {
			if (t < best.t || (t == best.t && best.obj && o.rank > best.obj->rank)) {
				best.t = t;
				best.obj = &o;
			}
		}
	}
}
// A node is "gated" when the box the reference computes for it differs from its true bounds.
// That happens only below a split that separates an inverted (negative-radius) sphere from
// the outer shell that would have covered it: the reference then reaches that inner sphere
// only through the per-axis test of src/math.rs:356-371 on the smaller box, and misses it
// for some rays.  TIGHT reproduces that by applying the reference's test on the reference's
// box in addition to the slab test on the true bounds.  For every other node the reference's
// test is implied by the slab test.
static inline bool is_gated(const BVH* n) {
	return n->aabb.min.x != n->tight.min.x || n->aabb.min.y != n->tight.min.y || n->aabb.min.z != n->tight.min.z ||
	       n->aabb.max.x != n->tight.max.x || n->aabb.max.y != n->tight.max.y || n->aabb.max.z != n->tight.max.z;
}
static void tight_visit(const BVH* n, const Ray& ray, Vec3 inv, TightBest& best, Counters& c) {
	if (n->is_leaf) { tight_leaf(n, ray, best, c); return; }
	Scalar el, er;
	c.box_tests += 2;
	bool hl = slab_test(n->left->tight, ray, inv, best.t, &el);
	bool hr = slab_test(n->right->tight, ray, inv, best.t, &er);
	if (hl && is_gated(n->left.get())) hl = aabb_hit_reference(n->left->aabb, ray);
	if (hr && is_gated(n->right.get())) hr = aabb_hit_reference(n->right->aabb, ray);
	const BVH* first = n->left.get();
	const BVH* second = n->right.get();
	Scalar e1 = el, e2 = er;
	bool h1 = hl, h2 = hr;
	if (hl && hr && er < el) {  // nearer child first; the left child on equal entry
		std::swap(first, second); std::swap(e1, e2); std::swap(h1, h2);
	}
	if (h1 && !(e1 > best.t * kSlabPad)) tight_visit(first, ray, inv, best, c);
	if (h2 && !(e2 > best.t * kSlabPad)) tight_visit(second, ray, inv, best, c);
}
static const Object* bvh_hit_tight(const BVH* root, const Ray& ray, Scalar* t_out, Counters& c) {
	TightBest best{DBL_MAX, nullptr};
	Vec3 inv = 1.0 / ray.dir;
	if (is_gated(root) && !aabb_hit_reference(root->aabb, ray)) {
		*t_out = best.t;
		return nullptr;
	}
	if (root->is_leaf) {
		tight_leaf(root, ray, best, c);
	} else {
		tight_visit(root, ray, inv, best, c);
	}
	*t_out = best.t;
	return best.obj;
}

enum { MODE_FAITHFUL = 0, MODE_TIGHT = 1 };

// bvh.rs:130-142
static inline bool bvh_trace(const BVH* bvh, const Ray& ray, int mode, RayHit* hit, Counters& c) {
	c.rays++;
	Scalar distance;
	const Object* object =
	    mode == MODE_FAITHFUL ? bvh_hit_faithful(bvh, ray, &distance, c) : bvh_hit_tight(bvh, ray, &distance, c);
	if (!object) return false;
	Vec3 p = ray.at(distance);
	Vec3 normal = object_normal(*object, p);
	bool front_face = dot(ray.dir, normal) < 0.0;
	if (!front_face) normal = -normal;
	hit->point = p;
	hit->normal = normal;
	hit->material = &object->material;
	hit->front_face = front_face;
	hit->object_id = object->id;
	hit->t = distance;
	return true;
}

// ---------------------------------------------------------------------------------------
// src/camera.rs
// ---------------------------------------------------------------------------------------
struct Camera {  // camera.rs:4-12
	Vec3 origin, pixel00_loc, pixel_delta_x, pixel_delta_y, defocus_disk_x, defocus_disk_y;
	Scalar defocus_angle;
};
// camera.rs:25-69
static Camera camera_new(Vec3 origin, Vec3 direction, Scalar fov, Scalar focus_dist, Scalar defocus_angle, size_t width,
                         size_t height) {
	const Vec3 WORLD_UP = V(0.0, 1.0, 0.0);
	Scalar theta = radians(fov);
	Scalar h = std::tan(theta / 2.0);
	Scalar viewport_height = 2.0 * h * focus_dist;
	Scalar viewport_width = viewport_height * ((Scalar)width / (Scalar)height);
	Vec3 w = -normalize(direction);
	Vec3 u = normalize(cross(WORLD_UP, w));
	Vec3 v = cross(w, u);
	Vec3 viewport_u = u * viewport_width;
	Vec3 viewport_v = (-v) * viewport_height;
	Vec3 pixel_delta_x = viewport_u / (Scalar)width;
	Vec3 pixel_delta_y = viewport_v / (Scalar)height;
	Vec3 viewport_upper_left = origin - (w * focus_dist) - viewport_u / 2.0 - viewport_v / 2.0;
	Vec3 pixel00_loc = viewport_upper_left + (pixel_delta_x + pixel_delta_y) * 0.5;
	Scalar defocus_radius = focus_dist * std::tan(radians(defocus_angle / 2.0));
	Vec3 defocus_disk_x = u * defocus_radius;
	Vec3 defocus_disk_y = v * defocus_radius;
	return Camera{origin, pixel00_loc, pixel_delta_x, pixel_delta_y, defocus_disk_x, defocus_disk_y, defocus_angle};
}
// camera.rs:95-103
static Vec3 get_camera_rotation(Scalar yaw, Scalar pitch) {
	Scalar pitch_radians = pitch * (M_PI / 180.0);  // f64::to_radians multiplies by PI/180
	Scalar yaw_radians = yaw * (M_PI / 180.0);
	return V(std::cos(yaw_radians) * std::cos(pitch_radians), std::sin(pitch_radians),
	         std::sin(yaw_radians) * std::cos(pitch_radians));
}
// camera.rs:83-92 with :72-75 and :78-81.  Draw order: jitter x, jitter y, then (only when
// defocus_angle > 0) disk x, disk y.  The defocus offset multiplies the two scaled basis
// vectors component-wise (camera.rs:80) — kept as written.
// jitter=false is the parity probe's deterministic variant: pixel centre, camera origin.
static inline Ray camera_get_ray(const Camera& cam, Scalar x, Scalar y, PathRng& rng, bool jitter) {
	Vec3 pixel_center = cam.pixel00_loc + (cam.pixel_delta_x * x) + (cam.pixel_delta_y * y);
	if (!jitter) return Ray{cam.origin, normalize(pixel_center - cam.origin)};
	Scalar jx = random_range(-0.5, 0.5, rng);
	Scalar jy = random_range(-0.5, 0.5, rng);
	Vec3 pixel_sample = pixel_center + ((jx * cam.pixel_delta_x) + (jy * cam.pixel_delta_y));
	Vec3 ray_origin;
	if (cam.defocus_angle <= 0.0) {
		ray_origin = cam.origin;
	} else {
		Vec3 p = random_in_unit_disk(rng);
		ray_origin = cam.origin + (p.x * cam.defocus_disk_x) * (p.y * cam.defocus_disk_y);
	}
	return Ray{ray_origin, normalize(pixel_sample - ray_origin)};
}

// ---------------------------------------------------------------------------------------
// src/renderer.rs
// ---------------------------------------------------------------------------------------
struct RenderCtx {
	const BVH* bvh;
	int mode;
	int sky_mode;
};
// renderer.rs:3-26, literally (FAITHFUL).  `contribution` already holds the product of the
// attenuations when the sky is reached, and the unwinding multiplies by them again, so the
// sky term is weighted by the squared throughput.  A metal hit that absorbs (None) returns
// zero and drops that hit's emission.
static Vec3 ray_color_recursive(const Ray& ray, const RenderCtx& rc, Vec3* contribution, int depth, PathRng& rng,
                                Counters& c) {
	if (depth <= 0) return V(0, 0, 0);
	RayHit hit;
	if (bvh_trace(rc.bvh, ray, rc.mode, &hit, c)) {
		Scattered s;
		if (scatter(*hit.material, ray, hit, rng, /*use_pow=*/true, &s)) {
			switch (hit.material->type) {
			case MAT_LAMBERTIAN: c.scatter_lambertian++; break;
			case MAT_METAL: c.scatter_metal++; break;
			default: c.scatter_dielectric++; break;
			}
			*contribution = (*contribution) * s.attenuation;
			return s.attenuation * ray_color_recursive(s.scattered, rc, contribution, depth - 1, rng, c) +
			       emission_color(*hit.material);
		}
		c.scatter_metal++;
		return V(0, 0, 0);
	}
	c.sky_misses++;
	return sky_color(ray.dir, rc.sky_mode) * (*contribution);
}
// The same estimator unrolled (TIGHT, and the form the GPU kernels use):
//   T = 1, L = 0; per bounce: miss -> L += T * (sky * T), stop; absorbed -> stop;
//   else L += T * emission, T *= attenuation.
static Vec3 ray_color_iterative(Ray ray, const RenderCtx& rc, int max_depth, PathRng& rng, Counters& c) {
	Vec3 T = V(1, 1, 1), L = V(0, 0, 0);
	for (int depth = max_depth; depth > 0; depth--) {
		RayHit hit;
		if (!bvh_trace(rc.bvh, ray, rc.mode, &hit, c)) {
			c.sky_misses++;
			L = L + T * (sky_color(ray.dir, rc.sky_mode) * T);
			break;
		}
		Scattered s;
		bool ok = scatter(*hit.material, ray, hit, rng, /*use_pow=*/false, &s);
		switch (hit.material->type) {
		case MAT_LAMBERTIAN: c.scatter_lambertian++; break;
		case MAT_METAL: c.scatter_metal++; break;
		default: c.scatter_dielectric++; break;
		}
		if (!ok) break;
		L = L + T * emission_color(*hit.material);
		T = T * s.attenuation;
		ray = s.scattered;
	}
	return L;
}
// renderer.rs:29-40
static Vec3 render_sample(Scalar x, Scalar y, const Camera& cam, const RenderCtx& rc, int max_depth, PathRng& rng,
                          Counters& c) {
	c.paths++;
	Ray ray = camera_get_ray(cam, x, y, rng, true);
	if (rc.mode == MODE_FAITHFUL) {
		Vec3 contribution = V(1, 1, 1);
		return ray_color_recursive(ray, rc, &contribution, max_depth, rng, c);
	}
	return ray_color_iterative(ray, rc, max_depth, rng, c);
}

// ---------------------------------------------------------------------------------------
// C ABI used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
// ---------------------------------------------------------------------------------------
extern "C" {

// Same layout as the product's rthip_object (include/rthip.h), restated here so that the
// oracle shares no header with the code it checks.
struct oracle_object {
	int32_t kind;      // 0 sphere, 1 cube
	int32_t material;  // 0 Lambertian, 1 Metal, 2 Dielectric
	double center[3];
	double extent[3];  // sphere: extent[0] = radius ; cube: half extents
	double albedo[3];
	double param;  // emission / fuzz / ir
};
struct oracle_camera_desc {
	double origin[3], direction[3];
	double fov, focus_dist, defocus_angle;
	int32_t width, height;
};
struct oracle_render_params {
	uint32_t spp_begin, spp_count;
	int32_t max_depth;
	uint32_t seed;
	int32_t sky_mode;
	int32_t mode;     // 0 faithful, 1 tight
	int32_t threads;  // <= 0: hardware_concurrency
	int32_t x0, y0, x1, y1;  // pixel window rendered (x1/y1 exclusive); 0,0,0,0 = whole image
};
struct oracle_scene {
	std::unique_ptr<BVH> bvh;
	int64_t n_objects;
};

oracle_scene* oracle_scene_create_ordered(const oracle_object* objs, int64_t n, int32_t tie_reverse);
oracle_scene* oracle_scene_create(const oracle_object* objs, int64_t n) { return oracle_scene_create_ordered(objs, n, 0); }
oracle_scene* oracle_scene_create_ordered(const oracle_object* objs, int64_t n, int32_t tie_reverse) {
	std::vector<Object> objects;
	objects.reserve((size_t)n);
	for (int64_t i = 0; i < n; i++) {
		Object o;
		o.kind = objs[i].kind;
		o.id = (int)i;
		o.center = V(objs[i].center[0], objs[i].center[1], objs[i].center[2]);
		o.ext = V(objs[i].extent[0], objs[i].extent[1], objs[i].extent[2]);
		o.material = Material{objs[i].material, V(objs[i].albedo[0], objs[i].albedo[1], objs[i].albedo[2]), objs[i].param};
		o.aabb = object_aabb(o.kind, o.center, o.ext);
		o.rank = 0;
		objects.push_back(o);
	}
	std::unique_ptr<BVH> bvh = bvh_new(std::move(objects), tie_reverse != 0);
	if (!bvh) return nullptr;  // bvh.rs:51 : an empty scene has no BVH
	int next = 0;
	assign_ranks(bvh.get(), &next);
	oracle_scene* s = new oracle_scene();
	s->bvh = std::move(bvh);
	s->n_objects = n;
	return s;
}
void oracle_scene_destroy(oracle_scene* s) { delete s; }
int32_t oracle_bvh_depth(const oracle_scene* s) { return bvh_depth(s->bvh.get()); }
// Object ids in leaf order (left to right) and the size of every leaf.
int64_t oracle_bvh_leaf_order(const oracle_scene* s, int32_t* ids, int32_t* leaf_sizes, int64_t* n_leaves) {
	std::vector<int> v, l;
	bvh_leaf_order(s->bvh.get(), &v, &l);
	if (ids) for (size_t i = 0; i < v.size(); i++) ids[i] = v[i];
	if (leaf_sizes) for (size_t i = 0; i < l.size(); i++) leaf_sizes[i] = l[i];
	if (n_leaves) *n_leaves = (int64_t)l.size();
	return (int64_t)v.size();
}

static Camera camera_from_desc(const oracle_camera_desc* d) {
	return camera_new(V(d->origin[0], d->origin[1], d->origin[2]), V(d->direction[0], d->direction[1], d->direction[2]),
	                  d->fov, d->focus_dist, d->defocus_angle, (size_t)d->width, (size_t)d->height);
}
// Derived camera fields in the order origin, pixel00, delta_x, delta_y, disk_x, disk_y (18
// values) followed by defocus_angle.
void oracle_camera_fields(const oracle_camera_desc* d, double out[19]) {
	Camera c = camera_from_desc(d);
	const Vec3* v[6] = {&c.origin, &c.pixel00_loc, &c.pixel_delta_x, &c.pixel_delta_y, &c.defocus_disk_x, &c.defocus_disk_y};
	for (int i = 0; i < 6; i++) { out[3 * i] = v[i]->x; out[3 * i + 1] = v[i]->y; out[3 * i + 2] = v[i]->z; }
	out[18] = c.defocus_angle;
}
void oracle_camera_rotation(double yaw, double pitch, double out[3]) {
	Vec3 r = get_camera_rotation(yaw, pitch);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

// Parity probe: one ray per pixel of the window.  jitter=0: through the pixel centre from
// the camera origin.  jitter=1: the ray sample `sample` of the render would start with.
// id = -1 and t = 0 on a miss.
void oracle_trace_primary(const oracle_scene* s, const oracle_camera_desc* cd, int32_t mode, int32_t jitter, uint32_t seed,
                          uint32_t sample, int32_t* id, double* t, double* normal, uint8_t* front) {
	Camera cam = camera_from_desc(cd);
	Counters c{};
	for (int y = 0; y < cd->height; y++) {
		for (int x = 0; x < cd->width; x++) {
			size_t i = (size_t)y * cd->width + x;
			PathRng rng;
			rng.init(seed, (uint32_t)i, sample);
			Ray ray = camera_get_ray(cam, (Scalar)x, (Scalar)y, rng, jitter != 0);
			RayHit hit;
			if (bvh_trace(s->bvh.get(), ray, mode, &hit, c)) {
				id[i] = hit.object_id; t[i] = hit.t;
				normal[3 * i] = hit.normal.x; normal[3 * i + 1] = hit.normal.y; normal[3 * i + 2] = hit.normal.z;
				front[i] = hit.front_face ? 1 : 0;
			} else {
				id[i] = -1; t[i] = 0.0;
				normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = 0.0;
				front[i] = 0;
			}
		}
	}
}

// The frame loop of raytracer_headless/src/main.rs:34-57: row-parallel, per pixel the sum
// over samples of sqrt(render()) (gamma per sample, before averaging).  `accum` (W*H*3) is
// ADDED to, the caller divides by the sample count (main.rs:55).  `samples_out`, when not
// null, receives the linear radiance of every sample, laid out [sample][y][x][3] over the
// whole image.  Returns the wall time of the loop in seconds.
double oracle_render(const oracle_scene* s, const oracle_camera_desc* cd, const oracle_render_params* p, double* accum,
                     double* samples_out, Counters* counters_out) {
	Camera cam = camera_from_desc(cd);
	RenderCtx rc{s->bvh.get(), p->mode, p->sky_mode};
	int W = cd->width, H = cd->height;
	int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
	if (x1 <= x0 || y1 <= y0) { x0 = 0; y0 = 0; x1 = W; y1 = H; }
	int nthreads = p->threads > 0 ? p->threads : (int)std::thread::hardware_concurrency();
	if (nthreads < 1) nthreads = 1;
	std::atomic<int> next_row(y0);
	std::vector<Counters> per_thread((size_t)nthreads, Counters{});
	auto t_start = std::chrono::steady_clock::now();
	auto worker = [&](int tid) {
		Counters c{};
		for (;;) {
			int y = next_row.fetch_add(1);  // one task per row, as par_chunks_exact_mut(width)
			if (y >= y1) break;
			for (int x = x0; x < x1; x++) {
				size_t i = (size_t)y * W + x;
				Vec3 final_color = V(0, 0, 0);
				for (uint32_t k = 0; k < p->spp_count; k++) {
					uint32_t sample = p->spp_begin + k;
					PathRng rng;
					rng.init(p->seed, (uint32_t)i, sample);
					Vec3 col = render_sample((Scalar)x, (Scalar)y, cam, rc, p->max_depth, rng, c);
					c.rng_draws += rng.draws;
					if (samples_out) {
						double* o = samples_out + (((size_t)k * H + y) * W + x) * 3;
						o[0] = col.x; o[1] = col.y; o[2] = col.z;
					}
					final_color = final_color + linear_to_gamma(col);
				}
				accum[3 * i] += final_color.x; accum[3 * i + 1] += final_color.y; accum[3 * i + 2] += final_color.z;
			}
		}
		per_thread[(size_t)tid] = c;
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < nthreads; t++) pool.emplace_back(worker, t);
	worker(0);
	for (auto& th : pool) th.join();
	double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	if (counters_out) {
		Counters total{};
		for (const Counters& c : per_thread) counters_add(total, c);
		*counters_out = total;
	}
	return secs;
}

// raytracer_headless/src/main.rs:6-12 : `(v * 255.0) as u8` — truncation toward zero,
// saturating at 0 and 255, NaN -> 0.  `mean` is accum / samples.
void oracle_quantize_rgb8(const double* mean, int64_t n_values, uint8_t* out) {
	for (int64_t i = 0; i < n_values; i++) {
		double v = mean[i] * 255.0;
		uint8_t q;
		if (std::isnan(v)) q = 0;
		else if (v <= 0.0) q = 0;
		else if (v >= 255.0) q = 255;
		else q = (uint8_t)v;
		out[i] = q;
	}
}

// ---- single-function probes for known-answer tests -------------------------------------
static Object probe_object(int kind, const double c[3], const double e[3]) {
	Object o{};
	o.kind = kind;
	o.center = V(c[0], c[1], c[2]);
	o.ext = V(e[0], e[1], e[2]);
	o.aabb = object_aabb(kind, o.center, o.ext);
	return o;
}
// Returns 1 and writes t when the reference's hit() is Some (before the t > 0.001 filter of
// scene.rs:105), else 0.
int32_t oracle_object_hit(int32_t kind, const double c[3], const double e[3], const double o[3], const double d[3], double* t) {
	Object ob = probe_object(kind, c, e);
	Ray r{V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
	Scalar tt = 0;
	bool h = object_hit(ob, r, &tt);
	*t = tt;
	return h ? 1 : 0;
}
void oracle_object_normal(int32_t kind, const double c[3], const double e[3], const double p[3], double n[3]) {
	Object ob = probe_object(kind, c, e);
	Vec3 r = object_normal(ob, V(p[0], p[1], p[2]));
	n[0] = r.x; n[1] = r.y; n[2] = r.z;
}
void oracle_object_aabb(int32_t kind, const double c[3], const double e[3], double mn[3], double mx[3]) {
	Object ob = probe_object(kind, c, e);
	mn[0] = ob.aabb.min.x; mn[1] = ob.aabb.min.y; mn[2] = ob.aabb.min.z;
	mx[0] = ob.aabb.max.x; mx[1] = ob.aabb.max.y; mx[2] = ob.aabb.max.z;
}
int32_t oracle_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3]) {
	AABB b{V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2])};
	Ray r{V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
	return aabb_hit_reference(b, r) ? 1 : 0;
}
int32_t oracle_slab_hit(const double mn[3], const double mx[3], const double o[3], const double d[3], double best_t, double* entry) {
	AABB b{V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2])};
	Ray r{V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
	return slab_test(b, r, 1.0 / r.dir, best_t, entry) ? 1 : 0;
}
void oracle_sky(const double d[3], int32_t sky_mode, double out[3]) {
	Vec3 c = sky_color(V(d[0], d[1], d[2]), sky_mode);
	out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
double oracle_reflectance(double cosine, double ref_idx, int32_t use_pow) { return reflectance(cosine, ref_idx, use_pow != 0); }
void oracle_reflect(const double v[3], const double n[3], double out[3]) {
	Vec3 r = reflect(V(v[0], v[1], v[2]), V(n[0], n[1], n[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_refract(const double uv[3], const double n[3], double eta, double out[3]) {
	Vec3 r = refract(V(uv[0], uv[1], uv[2]), V(n[0], n[1], n[2]), eta);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// One scatter event with the stream of (seed, pixel, sample).  Returns 1 for Some, 0 for None;
// writes attenuation, the scattered direction and the number of draws consumed.
int32_t oracle_scatter(int32_t mat_type, const double albedo[3], double param, const double dir_in[3], const double point[3],
                       const double normal[3], int32_t front_face, uint32_t seed, uint32_t pixel, uint32_t sample,
                       int32_t use_pow, double att[3], double dir_out[3], int32_t* draws) {
	Material m{mat_type, V(albedo[0], albedo[1], albedo[2]), param};
	Ray rin{V(0, 0, 0), V(dir_in[0], dir_in[1], dir_in[2])};
	RayHit hit{};
	hit.point = V(point[0], point[1], point[2]);
	hit.normal = V(normal[0], normal[1], normal[2]);
	hit.material = &m;
	hit.front_face = front_face != 0;
	PathRng rng;
	rng.init(seed, pixel, sample);
	Scattered s{};
	bool ok = scatter(m, rin, hit, rng, use_pow != 0, &s);
	*draws = (int32_t)rng.draws;
	if (ok) {
		att[0] = s.attenuation.x; att[1] = s.attenuation.y; att[2] = s.attenuation.z;
		dir_out[0] = s.scattered.dir.x; dir_out[1] = s.scattered.dir.y; dir_out[2] = s.scattered.dir.z;
	}
	return ok ? 1 : 0;
}
void oracle_emission(int32_t mat_type, const double albedo[3], double param, double out[3]) {
	Material m{mat_type, V(albedo[0], albedo[1], albedo[2]), param};
	Vec3 e = emission_color(m);
	out[0] = e.x; out[1] = e.y; out[2] = e.z;
}
// First n words of the stream of (seed, pixel, sample).
void oracle_rng_words(uint32_t seed, uint32_t pixel, uint32_t sample, int32_t n, uint32_t* out) {
	PathRng r;
	r.init(seed, pixel, sample);
	for (int i = 0; i < n; i++) out[i] = r.next_u32();
}
// The ray get_ray() returns for sample `sample` of pixel (x, y).
void oracle_camera_ray(const oracle_camera_desc* cd, int32_t x, int32_t y, uint32_t seed, uint32_t sample, int32_t jitter,
                       double origin[3], double dir[3]) {
	Camera cam = camera_from_desc(cd);
	PathRng rng;
	rng.init(seed, (uint32_t)((size_t)y * cd->width + x), sample);
	Ray r = camera_get_ray(cam, (Scalar)x, (Scalar)y, rng, jitter != 0);
	origin[0] = r.origin.x; origin[1] = r.origin.y; origin[2] = r.origin.z;
	dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

// The rays get_ray() returns for sample `sample` of every pixel, row-major [H][W][3].
void oracle_camera_rays(const oracle_camera_desc* cd, uint32_t seed, uint32_t sample, int32_t jitter, double* origins, double* dirs) {
	Camera cam = camera_from_desc(cd);
	for (int y = 0; y < cd->height; y++) {
		for (int x = 0; x < cd->width; x++) {
			size_t i = (size_t)y * cd->width + x;
			PathRng rng;
			rng.init(seed, (uint32_t)i, sample);
			Ray r = camera_get_ray(cam, (Scalar)x, (Scalar)y, rng, jitter != 0);
			origins[3 * i] = r.origin.x; origins[3 * i + 1] = r.origin.y; origins[3 * i + 2] = r.origin.z;
			dirs[3 * i] = r.dir.x; dirs[3 * i + 1] = r.dir.y; dirs[3 * i + 2] = r.dir.z;
		}
	}
}

}  // extern "C"
