// kernels.hpp — the path-tracing hot path as gfx950 device code, templated on the arithmetic
// type.  Instantiated twice: kernels_f32.hip (production; compiled with fp contraction) and
// kernels_f64.hip (the reference's own f64 arithmetic, operation for operation, compiled with
// -ffp-contract=off so that results can be compared bit for bit with a CPU restatement).
//
// What replaces what (reference file:line):
//   camera_ray        Camera::get_ray                     src/camera.rs:72-92
//   hit_sphere/_cube  Sphere::hit, Cube::hit              src/shapes/sphere.rs:26-48, cube.rs:24-42
//   Traversal         BVH::hit + Scene::hit               src/bvh.rs:84-127, src/scene.rs:100-113
//   make_hit          BVH::trace, get_normal              src/bvh.rs:130-142, sphere.rs:60-62, cube.rs:54-64
//   scatter           Material::scatter / reflectance     src/materials.rs:17-85
//   sky               Scene::get_sky_color                src/scene.rs:116-120
//   render_kernel     render() + ray_color() + the sample loop
//                                                         src/renderer.rs:3-40, raytracer_headless/src/main.rs:41-55
//
// Execution model.  One launch renders `spp_count` samples of every pixel the call owns.  The
// grid is persistent: every wave pulls work items (an 8x8 tile x `chunk` samples = 64*chunk
// paths) from one of eight counters (one per range of the image, so that the workgroups that
// share an XCD's L2 work on one band) and keeps its 64 lanes busy by handing finished lanes the
// next paths of the wave's pool (ballot + mbcnt ranking), so lanes are not tied to pixels.  A
// lane is a small state machine: IDLE -> FRESH -> TRAVERSE -> SHADE -> (FRESH | IDLE).  The wave
// alternates between shading, regeneration (once enough lanes are idle), ray set-up (which
// includes the test of the scene's oversized primitives for all starting lanes together), and a
// traversal loop, which it leaves once the number of lanes still traversing drops under a
// threshold while the other steps have lanes to serve; lanes interrupted mid-traversal keep
// their stack and simply continue on the next round.  The per-lane
// traversal stack lives in LDS ([entry][thread], conflict-free); scenes that fit are copied to
// LDS as well, so the inner loop touches no global memory.  Finished paths add sqrt(radiance)
// to the accumulation buffer as 64-bit fixed-point integers (atomic adds commute exactly, so
// the image does not depend on scheduling, on the tile partition or on the GPU count).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "layout.hpp"

namespace rthip {

#ifndef RT_QWAVES
#define RT_QWAVES 6   // (7, i.e. 72 registers, before the camera-ray batches and their packet walk came to these kernels)
#endif
#ifndef RT_MINWAVES
#define RT_MINWAVES 6
#endif
#ifndef RT_ROOTWAVES
#define RT_ROOTWAVES 7   // the kernel of one-leaf scenes (no traversal code): 68 registers
#endif
constexpr uint32_t kQueues = 8;   // work-item ranges, one per XCD of the part
constexpr int kProbeBlock = 256;  // threads per workgroup of the parity probe
constexpr int kDone = INT32_MIN;  // traversal finished (not a valid leaf code)

template <typename Real>
struct RenderArgs {
	const char* scene;  // [nodes | prims | gates | mats] blob in HBM, each part 64-byte aligned
	uint32_t prims_off, mats_off, gates_off, scene_bytes;
	uint32_t acc_tile_off;  // kAccLds: byte offset in LDS of the waves' accumulation tiles (1536 bytes per wave)
	uint32_t lds_bytes;     // leading part of the blob that LDS-resident launches copy: nodes, primitives, gates, and (kLds == 2) the materials
	uint32_t qnodes_off;    // quantised nodes (f32 blobs of large scenes), 0 = none
	QFrame qframe;          // their grid
	int32_t big_leaf;       // leaf code of the oversized primitives every ray tests first, 0 = none
	int32_t start;          // where traversal starts after them: an inner node or a leaf code (0 without big_leaf)
	int32_t stack_entries;  // tree depth
	CameraT<Real> cam;
	int32_t width, height, tiles_x, tiles_y;
	uint32_t spp_begin, spp_count;
	int32_t max_depth;
	uint32_t seed;
	int32_t sky_mode;
	int32_t part_index, part_count;
	uint32_t n_items, chunk, n_chunks;
	int32_t trav_threshold;
	uint32_t block_order;   // S > 0: work items run through the image in blocks of S x S tiles (scenes in global memory)
	int32_t regen_min;      // regenerate paths only when at least this many lanes are idle (or all are)
	int32_t inner_exit;     // sixteenths: leave the inner-node loop when fewer lanes than this share of the round remain in it
	unsigned long long* accum;    // [tiles][3 channels][64 px] fixed point
	Real* samples_out;            // optional [spp_count][H][W][3]
	uint32_t* queue;              // kQueues work-item counters, one per 64-byte line, zeroed before the launch
	unsigned long long* stats;    // [0] paths [1] rays [2] box tests [3] sphere tests [4] cube tests, then region counters
	char* stack_mem;              // kStackMem: the lanes' traversal stacks, stack_entries x (grid x threads) words
	uint32_t xchg_off, xchg_cap;  // kXchg: byte offset in LDS of the workgroup's exchange queues, and records per queue (a power of two)
	int32_t xchg_pop_min;         // kXchg: take rays out of a queue only this many at a time (or when the wave has nothing else to do)
	int32_t xchg_push_max;        // kXchg: an arm hands its rays over only when it holds at most this many lanes of the wave
	int32_t packet;               // camera rays are traced 64 at a time as one packet per wave (kernels that make them in batches)
	int32_t pixel_major;          // such batches run a work item pixel by pixel (a packet = all samples of a few neighbouring pixels) instead of sample by sample
};

template <typename Real>
struct ProbeArgs {
	const char* scene;
	uint32_t prims_off, mats_off, gates_off, scene_bytes;
	int32_t stack_entries;
	CameraT<Real> cam;
	int32_t width, height;
	uint32_t seed, sample;
	int32_t jitter;
	int32_t* id;
	double* t;
	double* normal;
	uint8_t* front;
	// f32 probe only: the f64 camera and the f64 primitive records (same indices as the f32 ones).
	// With scene64 set, the primitive the f32 traversal found is intersected once more in f64 from
	// the f64 camera ray, and t, normal and front face are reported from that.
	CameraT<double> cam64;
	const char* scene64;
	uint32_t prims_off64;
};

// ---------------------------------------------------------------------------------------
// Arithmetic policy.  f64 follows the reference literally.  f32 uses the hardware's 1-ulp
// reciprocal / square root / reciprocal square root.
// ---------------------------------------------------------------------------------------
template <typename Real> struct Fp;
template <> struct Fp<float> {
	static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
	static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
	static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
	static __device__ __forceinline__ float uniform(uint32_t r) { return (float)(r >> 8) * 0x1p-24f; }
	static __device__ __forceinline__ float fmin(float a, float b) { return __builtin_fminf(a, b); }
	static __device__ __forceinline__ float fmax(float a, float b) { return __builtin_fmaxf(a, b); }
	static __device__ __forceinline__ float fabs(float a) { return __builtin_fabsf(a); }
	static __device__ __forceinline__ float copysign(float a, float b) { return __builtin_copysignf(a, b); }
	static __device__ __forceinline__ float inf() { return __builtin_inff(); }
	static __device__ __forceinline__ float max_finite() { return 3.402823466e+38f; }
};
template <> struct Fp<double> {
	static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
	static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
	static __device__ __forceinline__ double uniform(uint32_t r) { return (double)r * (1.0 / 4294967296.0); }
	static __device__ __forceinline__ double fmin(double a, double b) { return __builtin_fmin(a, b); }
	static __device__ __forceinline__ double fmax(double a, double b) { return __builtin_fmax(a, b); }
	static __device__ __forceinline__ double fabs(double a) { return __builtin_fabs(a); }
	static __device__ __forceinline__ double copysign(double a, double b) { return __builtin_copysign(a, b); }
	static __device__ __forceinline__ double inf() { return __builtin_inf(); }
	static __device__ __forceinline__ double max_finite() { return 1.7976931348623157e+308; }
};

// src/math.rs:18-318
template <typename Real> struct V3 { Real x, y, z; };
#define RT_FN template <typename Real> __device__ __forceinline__
RT_FN V3<Real> mk(Real x, Real y, Real z) { return V3<Real>{x, y, z}; }
RT_FN V3<Real> ld3(const Real* p) { return V3<Real>{p[0], p[1], p[2]}; }
RT_FN V3<Real> operator+(V3<Real> a, V3<Real> b) { return mk<Real>(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_FN V3<Real> operator-(V3<Real> a, V3<Real> b) { return mk<Real>(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_FN V3<Real> operator-(V3<Real> a) { return mk<Real>(-a.x, -a.y, -a.z); }
RT_FN V3<Real> operator*(V3<Real> a, Real s) { return mk<Real>(a.x * s, a.y * s, a.z * s); }
RT_FN V3<Real> operator*(Real s, V3<Real> a) { return mk<Real>(a.x * s, a.y * s, a.z * s); }
RT_FN V3<Real> operator*(V3<Real> a, V3<Real> b) { return mk<Real>(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_FN V3<Real> operator/(V3<Real> a, Real s) { return mk<Real>(a.x / s, a.y / s, a.z / s); }
RT_FN Real dot(V3<Real> a, V3<Real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_FN Real length_squared(V3<Real> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// src/math.rs:78-85: divide by the length (f64); f32 multiplies by the reciprocal square root.
RT_FN V3<Real> normalize(V3<Real> a) {
	if constexpr (std::is_same<Real, float>::value) {
		return a * Fp<float>::rsqrt(length_squared(a));
	} else {
		Real l = Fp<Real>::sqrt(length_squared(a));
		return mk<Real>(a.x / l, a.y / l, a.z / l);
	}
}
// src/math.rs:97-99: v - ((n * 2) * dot(v, n))
RT_FN V3<Real> reflect(V3<Real> v, V3<Real> n) { return v - ((n * Real(2)) * dot(v, n)); }
// src/math.rs:102-107
RT_FN V3<Real> refract(V3<Real> uv, V3<Real> n, Real eta) {
	Real cos_theta = Fp<Real>::fmin(dot(-uv, n), Real(1));
	V3<Real> perp = eta * (uv + cos_theta * n);
	V3<Real> par = -(Fp<Real>::sqrt(Fp<Real>::fabs(Real(1) - length_squared(perp)))) * n;
	return perp + par;
}

// ---------------------------------------------------------------------------------------
// Counter-based random stream: PCG RXS-M-XS 32 whose state and increment are hashed from
// (seed, pixel index, sample index).  One draw is one 32-bit word; f32 uses its top 24 bits,
// f64 all 32, so both precisions walk the same stream.  The reference's generator is
// OS-seeded (src/math.rs:13-15, raytracer_headless/src/main.rs:40): only distributions and
// draw counts are contractual.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
struct Rng {
	uint32_t state, inc;
	__device__ __forceinline__ void init(uint32_t seed, uint32_t pixel, uint32_t sample) {
		uint32_t a = mix32(seed + 0x9E3779B9u * (pixel + 1u));
		uint32_t b = mix32(a ^ (0x85EBCA6Bu * (sample + 1u)));
		state = b;
		inc = (mix32(b + a + 0xC2B2AE35u) << 1) | 1u;
	}
	__device__ __forceinline__ void skip() { state = state * 747796405u + inc; }  // a draw whose value nobody looks at
	__device__ __forceinline__ uint32_t next() {
		uint32_t old = state;
		state = old * 747796405u + inc;
		uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
		return (word >> 22u) ^ word;
	}
};
// src/math.rs:13-15 random(min, max): half-open, uniform * scale + low
RT_FN Real random_range(Real lo, Real hi, Rng& r) { return lo + (hi - lo) * Fp<Real>::uniform(r.next()); }
// src/math.rs:156-168: three draws x, y, z in [-1, 1), normalised (a cube sample pushed to
// the sphere, not uniform on it: kept as written).
RT_FN V3<Real> random_unit_vector(Rng& r) {
	Real x = random_range<Real>(Real(-1), Real(1), r);
	Real y = random_range<Real>(Real(-1), Real(1), r);
	Real z = random_range<Real>(Real(-1), Real(1), r);
	return normalize(mk<Real>(x, y, z));
}

template <typename Real> struct RayT {
	V3<Real> o, d;
};

// src/camera.rs:83-92 (+ :72-81).  Draws: jitter x, jitter y, then disk x, disk y only when
// defocus_angle > 0.  The disk sample is normalised onto the unit circle (src/math.rs:171-174)
// and the two scaled basis vectors are MULTIPLIED component-wise (src/camera.rs:80): both are
// the reference's behaviour and are kept.
RT_FN RayT<Real> camera_ray(const CameraT<Real>& c, Real x, Real y, Rng& rng, bool jitter) {
	V3<Real> origin = ld3(c.origin), dx = ld3(c.delta_x), dy = ld3(c.delta_y);
	V3<Real> pixel_center = ld3(c.pixel00) + (dx * x) + (dy * y);
	if (!jitter) return RayT<Real>{origin, normalize(pixel_center - origin)};
	Real jx = random_range<Real>(Real(-0.5), Real(0.5), rng);
	Real jy = random_range<Real>(Real(-0.5), Real(0.5), rng);
	V3<Real> pixel_sample = pixel_center + ((jx * dx) + (jy * dy));
	V3<Real> ray_origin = origin;
	if (c.defocus_angle > Real(0)) {
		Real px = random_range<Real>(Real(-1), Real(1), rng);
		Real py = random_range<Real>(Real(-1), Real(1), rng);
		V3<Real> p = normalize(mk<Real>(px, py, Real(0)));
		ray_origin = origin + (p.x * ld3(c.disk_x)) * (p.y * ld3(c.disk_y));
	}
	return RayT<Real>{ray_origin, normalize(pixel_sample - ray_origin)};
}

// src/scene.rs:116-120, and the wallpaper sky (formula of
// raytracer_gpu/src/shaders/compute.wgsl:186-195 on the normalised direction).
RT_FN V3<Real> sky(V3<Real> u, int sky_mode) {  // u = the normalised direction
	if (sky_mode == RTHIP_SKY_WALLPAPER) {
		Real strength = Real(0.5) * (-u.y + Real(0.25));
		Real t = Real(0.5) * (u.x + Real(1));
		Real e = Real(2) * t - Real(1);
		Real pw;
		if constexpr (std::is_same<Real, float>::value) pw = __builtin_exp2f(6.62935662f * (e * e - 1.0f));  // 99^x
		else pw = ::pow(99.0, e * e - 1.0);
		strength *= Real(50) * pw;
		V3<Real> a = mk<Real>(Real(0.94), Real(0.02), Real(0.99)), b = mk<Real>(Real(0), Real(0.85), Real(0.98));
		V3<Real> c = mk<Real>(Real(0), Real(0.45), Real(0.98)), d = mk<Real>(Real(0), Real(0.98), Real(0.45));
		return (a * (Real(1) - t) + b * t) * strength + (Real(0.25) * c) * (Real(0.5) * (u.y + Real(1))) +
		       (Real(0.15) * d) * (Real(0.5) * (-u.x + Real(1)));
	}
	Real a = Real(0.5) * (u.y + Real(1));
	return mk<Real>(Real(1), Real(1), Real(1)) * (Real(1) - a) + mk<Real>(Real(0.5), Real(0.7), Real(1)) * a;
}

// ---------------------------------------------------------------------------------------
// A ray prepared for traversal.
// ---------------------------------------------------------------------------------------
template <typename Real> struct TravRay {
	V3<Real> o, d, inv;  // inv = 1 / d
	V3<Real> noi;        // -(o * inv): the f32 box test works on coordinate * inv + noi, one fma
	Real eps;            // absolute slack that covers the rounding of that fma form
	Real rcp_a;          // 1 / d.d (f32 sphere test)
	V3<Real> qscale;     // quantised nodes only: step / d (noi and eps are then relative to the grid centre)
	V3<Real> qneg;       // min(qscale, 0) per axis
};
// `frame` (f32 traversal over quantised nodes) moves the box-test operands into grid units; the
// primitive tests keep using o, d, inv.
RT_FN V3<Real> ray_inv(V3<Real> d) {
	V3<Real> inv = mk<Real>(Fp<Real>::rcp(d.x), Fp<Real>::rcp(d.y), Fp<Real>::rcp(d.z));
	if constexpr (std::is_same<Real, float>::value) {
		// A zero (or flushed) direction component gives inv = inf and plane * inf - o * inf = NaN for
		// every plane; a huge finite stand-in keeps the sign of (plane - o), which is all that axis
		// contributes for such a ray.
		inv.x = __builtin_copysignf(__builtin_fminf(__builtin_fabsf(inv.x), 0x1p64f), inv.x);
		inv.y = __builtin_copysignf(__builtin_fminf(__builtin_fabsf(inv.y), 0x1p64f), inv.y);
		inv.z = __builtin_copysignf(__builtin_fminf(__builtin_fabsf(inv.z), 0x1p64f), inv.z);
	}
	return inv;
}
// Kernels that walk quantised nodes do not keep inv and rcp_a between the primitive tests (the box
// test there works on qscale, and registers are what limits those kernels): the leaf step makes
// them again, by the formulas of prepare(), so the primitive tests see the same numbers.
template <bool kLean, typename Real>
__device__ __forceinline__ TravRay<Real> leaf_ray(const TravRay<Real>& r) {
	if constexpr (!kLean) return r;
	else {
		TravRay<Real> t = r;
		t.inv = ray_inv<Real>(r.d);
		t.rcp_a = Fp<Real>::rcp(dot(r.d, r.d));
		return t;
	}
}
RT_FN TravRay<Real> prepare(RayT<Real> r, const QFrame* frame = nullptr) {
	TravRay<Real> t;
	t.o = r.o; t.d = r.d;
	t.inv = ray_inv<Real>(r.d);
	V3<Real> box_o = r.o;
	t.qscale = t.inv;
	Real reach = Real(0);  // largest |plane| the box test multiplies by inv, beside the origin
	if constexpr (std::is_same<Real, float>::value) {
		if (frame) {
			box_o = r.o - ld3(frame->centre);
			t.qscale = ld3(frame->step) * t.inv;
			reach = Real(1);
		}
	}
	t.qneg = mk<Real>(Fp<Real>::fmin(t.qscale.x, Real(0)), Fp<Real>::fmin(t.qscale.y, Real(0)), Fp<Real>::fmin(t.qscale.z, Real(0)));
	t.noi = -(box_o * t.inv);
	// plane*inv - o*inv is exact to an ulp of |o*inv| (not of the result): widen by 4 ulps of the
	// largest term.  Only axes the ray actually advances along count (|d_a| >= 2^-12 max|d|): on a
	// nearly parallel axis that ulp, however many time units it is, is still one ulp of o_a in
	// space, and it can only drop a box whose face the ray skims within that distance; counting it
	// would make the slack exceed every box of the scene and the ray visit the whole tree.
	Real ax = Fp<Real>::fabs(r.d.x), ay = Fp<Real>::fabs(r.d.y), az = Fp<Real>::fabs(r.d.z);
	Real floor_d = Fp<Real>::fmax(Fp<Real>::fmax(ax, ay), az) * Real(0x1p-12);
	// quantised planes reach kQMax steps from the centre: q * qscale rounds like a term of that size
	Real ex = ax >= floor_d ? Fp<Real>::fmax(Fp<Real>::fabs(t.noi.x), reach * Real(kQMax) * Fp<Real>::fabs(t.qscale.x)) : Real(0);
	Real ey = ay >= floor_d ? Fp<Real>::fmax(Fp<Real>::fabs(t.noi.y), reach * Real(kQMax) * Fp<Real>::fabs(t.qscale.y)) : Real(0);
	Real ez = az >= floor_d ? Fp<Real>::fmax(Fp<Real>::fabs(t.noi.z), reach * Real(kQMax) * Fp<Real>::fabs(t.qscale.z)) : Real(0);
	t.eps = Fp<Real>::fmax(Fp<Real>::fmax(ex, ey), ez) * Real(0x1p-21);
	t.rcp_a = Fp<Real>::rcp(dot(r.d, r.d));
	return t;
}

// Sphere::hit, src/shapes/sphere.rs:26-48: the NEAR root only, direction not assumed unit.
// f64: the reference's quadratic, literally.  f32: the same root written as
//   t = -(oc.d)/a - sqrt((r^2 - |oc - ((oc.d)/a) d|^2) / a)
// which keeps the discriminant accurate when |oc| >> r (small spheres far away, or the sun
// sphere of radius 7500): b^2 - 4ac = 4a (r^2 - |q|^2) with q the closest approach to the centre.
RT_FN bool hit_sphere(const Prim<Real>& s, const TravRay<Real>& r, Real* t) {
	V3<Real> oc;
	Real radius;
	if constexpr (std::is_same<Real, float>::value) {
		const float4 cr = *reinterpret_cast<const float4*>(s.v);  // centre and radius in one 16-byte load (records are 32-byte aligned)
		oc = r.o - mk<Real>(cr.x, cr.y, cr.z);
		radius = cr.w;
	} else {
		oc = r.o - ld3(s.v);
		radius = s.v[3];
	}
	if constexpr (std::is_same<Real, float>::value) {
		Real tc = dot(oc, r.d) * r.rcp_a;  // -(parameter of the closest approach)
		V3<Real> q = oc - tc * r.d;
		Real disc = radius * radius - dot(q, q);
		if (disc < 0.0f) return false;
		*t = -tc - Fp<float>::sqrt(disc * r.rcp_a);
		return true;
	} else {
		Real a = dot(r.d, r.d);
		Real b = 2.0 * dot(oc, r.d);
		Real c = dot(oc, oc) - radius * radius;
		Real discriminant = b * b - 4.0 * a * c;
		if (discriminant < 0.0) return false;
		*t = (-b - Fp<double>::sqrt(discriminant)) / (2.0 * a);
		return true;
	}
}
// Cube::hit, src/shapes/cube.rs:24-42: entry distance from OUTSIDE only (t_n <= 0 -> None),
// a negative half extent never hits.  f64: the reference's centre/half-extent form.  f32: the
// same planes stored as lo = c - he, hi = c + he (rounded once from f64): per axis the near
// plane is lo when d >= 0 else hi, which is -n - k of the reference in exact arithmetic,
// without the cancellation of (o - c) against he for slabs of half extent 1e3..1e5.  The
// separately rounded (plane - o) * inv keeps t accurate to an ulp far from the origin, which
// the fused plane * inv - o * inv of the box test does not (measured: the fused form is also
// no faster here).
RT_FN bool hit_cube(const Prim<Real>& c, const TravRay<Real>& r, Real* t) {
	Real t_n, t_f;
	if constexpr (std::is_same<Real, float>::value) {
		static_assert(kBoxLoX == 0 && kBoxLoY == 1 && kBoxLoZ == 2 && kBoxHiX == 3 && kBoxHiY == 4 && kBoxHiZ == 5, "plane order of the loads below");
		const float4 p0 = *reinterpret_cast<const float4*>(c.v);      // the six planes in a 16-byte and an 8-byte load
		const float2 p1 = *reinterpret_cast<const float2*>(c.v + 4);
		float ax = (p0.x - r.o.x) * r.inv.x, bx = (p0.w - r.o.x) * r.inv.x;
		float ay = (p0.y - r.o.y) * r.inv.y, by = (p1.x - r.o.y) * r.inv.y;
		float az = (p0.z - r.o.z) * r.inv.z, bz = (p1.y - r.o.z) * r.inv.z;
		bool nx = r.inv.x < 0.0f, ny = r.inv.y < 0.0f, nz = r.inv.z < 0.0f;
		t_n = Fp<float>::fmax(Fp<float>::fmax(nx ? bx : ax, ny ? by : ay), nz ? bz : az);
		t_f = Fp<float>::fmin(Fp<float>::fmin(nx ? ax : bx, ny ? ay : by), nz ? az : bz);
	} else {
		V3<Real> origin = r.o - ld3(c.v);
		V3<Real> m = mk<Real>(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
		V3<Real> n = m * origin;
		V3<Real> k = mk<Real>(Fp<Real>::fabs(m.x), Fp<Real>::fabs(m.y), Fp<Real>::fabs(m.z)) * ld3(c.v + 3);
		V3<Real> t1 = -n - k;
		V3<Real> t2 = -n + k;
		t_n = Fp<Real>::fmax(Fp<Real>::fmax(t1.x, t1.y), t1.z);
		t_f = Fp<Real>::fmin(Fp<Real>::fmin(t2.x, t2.y), t2.z);
	}
	if (t_n > t_f || t_f <= Real(0) || t_n <= Real(0)) return false;
	*t = t_n;
	return true;
}

// True slab test of one child box against (0.001, best_t].  The box distances and a primitive's
// own t round differently (a cube's face and the face of the box around it are one plane
// computed by two formulas), so the far side is widened: culling stays conservative, and two
// coplanar faces of different leaves still meet in the (t, rank) comparison.
// f64: (plane - o) * inv per plane, relative pad 2^-30 (covers the cancellation of the
// reference's centre/half-extent cube formula for slabs of half extent 1e5).
// f32: one fma per plane (plane * inv - o * inv) and both interval ends folded into the
// min3/max3 chains; its rounding is absolute (an ulp of |o * inv|), so the far side gets the
// relative pad 2^-18 plus the ray's absolute slack eps (see prepare()).  A NaN plane distance
// drops out of min/max, which only makes the test more permissive.
// The fused f32 form on explicit operands: t = plane * scale + offset per axis.  For full nodes
// scale = 1/d and offset = -o/d; for quantised nodes the planes are grid steps and
// scale = step/d, offset = -(o - centre)/d.
__device__ __forceinline__ bool hit_box_fused(const float* b, V3<float> scale, V3<float> offset, float eps, float best_t, float* entry) {
	// One v_fma_f32 per plane.  (The two-wide v_pk_fma_f32 halves the instruction count here but
	// measured 2-4 % slower in this divergent loop.)
	float x0 = __builtin_fmaf(b[kBoxLoX], scale.x, offset.x), x1 = __builtin_fmaf(b[kBoxHiX], scale.x, offset.x);
	float y0 = __builtin_fmaf(b[kBoxLoY], scale.y, offset.y), y1 = __builtin_fmaf(b[kBoxHiY], scale.y, offset.y);
	float z0 = __builtin_fmaf(b[kBoxLoZ], scale.z, offset.z), z1 = __builtin_fmaf(b[kBoxHiZ], scale.z, offset.z);
	float tnear = Fp<float>::fmax(Fp<float>::fmax(Fp<float>::fmin(x0, x1), Fp<float>::fmin(y0, y1)), Fp<float>::fmin(z0, z1));
	float tfar = Fp<float>::fmin(Fp<float>::fmin(Fp<float>::fmax(x0, x1), Fp<float>::fmax(y0, y1)), Fp<float>::fmax(z0, z1));
	// min(tfar, best_t) on the bit patterns: best_t is positive (or +inf) and tfar is never NaN
	// (finite planes, finite scale and offset), so the signed-integer order is the float order,
	// and unlike v_min_f32 the integer min needs no canonicalised operand (the compiler re-did
	// v_max best_t, best_t in every iteration of the node loop).
	tfar = __builtin_bit_cast(float, __builtin_bit_cast(int32_t, tfar) < __builtin_bit_cast(int32_t, best_t) ? __builtin_bit_cast(int32_t, tfar) : __builtin_bit_cast(int32_t, best_t));
	*entry = tnear;
	return Fp<float>::fmax(tnear, 0.0009f) <= __builtin_fmaf(tfar, 1.0f + 0x1p-18f, eps);
}
// The f32 box of a full node is stored as centre and half extent: the distances to the two planes
// of an axis are tc -/+ th with tc = centre * scale + offset and th = half * |scale| (the abs is
// a source modifier), i.e. three FMAs per axis, and the near one is known without a min/max
// pair.  On this part v_fma/v_mul/v_add issue in 2-3 cycles per wave and v_min/v_max/v_cndmask
// in 4 (tools/microbench/oprate.hip), so this form is a third cheaper than two FMAs plus min and
// max per axis.  The host inflates the half extent by 2^-20 for the rounding of tc -/+ th.
__device__ __forceinline__ bool hit_box_centred(const float* b, V3<float> scale, V3<float> offset, float eps, float best_t, float* entry) {
	const float cx = __builtin_fmaf(b[kBoxLoX], scale.x, offset.x), ax = __builtin_fabsf(scale.x);
	const float cy = __builtin_fmaf(b[kBoxLoY], scale.y, offset.y), ay = __builtin_fabsf(scale.y);
	const float cz = __builtin_fmaf(b[kBoxLoZ], scale.z, offset.z), az = __builtin_fabsf(scale.z);
	const float tnear = Fp<float>::fmax(Fp<float>::fmax(__builtin_fmaf(-b[kBoxHiX], ax, cx), __builtin_fmaf(-b[kBoxHiY], ay, cy)), __builtin_fmaf(-b[kBoxHiZ], az, cz));
	float tfar = Fp<float>::fmin(Fp<float>::fmin(__builtin_fmaf(b[kBoxHiX], ax, cx), __builtin_fmaf(b[kBoxHiY], ay, cy)), __builtin_fmaf(b[kBoxHiZ], az, cz));
	tfar = __builtin_bit_cast(float, __builtin_bit_cast(int32_t, tfar) < __builtin_bit_cast(int32_t, best_t) ? __builtin_bit_cast(int32_t, tfar) : __builtin_bit_cast(int32_t, best_t));  // see hit_box_fused
	*entry = tnear;
	return Fp<float>::fmax(tnear, 0.0009f) <= __builtin_fmaf(tfar, 1.0f + 0x1p-18f, eps);
}
// Quantised child box: low plane and width per axis in grid steps.  With tlo the distance to the
// low plane, the near one is tlo + width * min(scale, 0) and the far one that + width * |scale|:
// three FMAs per axis, no min/max pair (see hit_box_centred for why that is cheaper here).
__device__ __forceinline__ bool hit_box_q(const float* b, V3<float> scale, V3<float> sneg, V3<float> offset, float eps, float best_t, float* entry) {
	const float nx = __builtin_fmaf(b[kBoxHiX], sneg.x, __builtin_fmaf(b[kBoxLoX], scale.x, offset.x));
	const float ny = __builtin_fmaf(b[kBoxHiY], sneg.y, __builtin_fmaf(b[kBoxLoY], scale.y, offset.y));
	const float nz = __builtin_fmaf(b[kBoxHiZ], sneg.z, __builtin_fmaf(b[kBoxLoZ], scale.z, offset.z));
	const float tnear = Fp<float>::fmax(Fp<float>::fmax(nx, ny), nz);
	float tfar = Fp<float>::fmin(Fp<float>::fmin(__builtin_fmaf(b[kBoxHiX], __builtin_fabsf(scale.x), nx), __builtin_fmaf(b[kBoxHiY], __builtin_fabsf(scale.y), ny)),
	                             __builtin_fmaf(b[kBoxHiZ], __builtin_fabsf(scale.z), nz));
	tfar = __builtin_bit_cast(float, __builtin_bit_cast(int32_t, tfar) < __builtin_bit_cast(int32_t, best_t) ? __builtin_bit_cast(int32_t, tfar) : __builtin_bit_cast(int32_t, best_t));  // see hit_box_fused
	*entry = tnear;
	return Fp<float>::fmax(tnear, 0.0009f) <= __builtin_fmaf(tfar, 1.0f + 0x1p-18f, eps);
}
RT_FN bool hit_box(const Real* b, const TravRay<Real>& r, Real best_t, Real* entry) {
	if constexpr (std::is_same<Real, float>::value) {
		return hit_box_centred(b, r.inv, r.noi, r.eps, best_t, entry);
	} else {
		Real tx0 = (b[kBoxLoX] - r.o.x) * r.inv.x, tx1 = (b[kBoxHiX] - r.o.x) * r.inv.x;
		Real ty0 = (b[kBoxLoY] - r.o.y) * r.inv.y, ty1 = (b[kBoxHiY] - r.o.y) * r.inv.y;
		Real tz0 = (b[kBoxLoZ] - r.o.z) * r.inv.z, tz1 = (b[kBoxHiZ] - r.o.z) * r.inv.z;
		Real tnear = Fp<Real>::fmax(Fp<Real>::fmax(Fp<Real>::fmin(tx0, tx1), Fp<Real>::fmin(ty0, ty1)), Fp<Real>::fmin(tz0, tz1));
		Real tfar = Fp<Real>::fmin(Fp<Real>::fmin(Fp<Real>::fmax(tx0, tx1), Fp<Real>::fmax(ty0, ty1)), Fp<Real>::fmax(tz0, tz1));
		*entry = tnear;
		const Real pad = Real(1.0 + 0x1p-30);
		tfar = tfar * pad;
		return tnear <= tfar && tfar > Real(0.001) && tnear <= best_t * pad;
	}
}

// AABB::hit, src/math.rs:356-371, as written: every axis is clipped against the constants
// [0.001, MAX] on its own (the interval is not carried from axis to axis).  Only used on the
// gate boxes of inverted spheres (see Gate in layout.hpp).
RT_FN bool hit_box_reference(const Real* b, const TravRay<Real>& r) {
	const Real kMax = std::is_same<Real, float>::value ? Real(3.402823466e+38) : Real(1.7976931348623157e+308);
	const Real o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
	bool ok = true;
#pragma unroll
	for (int a = 0; a < 3; a++) {
		Real t0 = (b[a] - o[a]) * inv[a];
		Real t1 = (b[3 + a] - o[a]) * inv[a];
		if (inv[a] < Real(0)) { Real s = t0; t0 = t1; t1 = s; }
		Real mn = t0 > Real(0.001) ? t0 : Real(0.001);
		Real mx = t1 < kMax ? t1 : kMax;
		ok = ok && !(mx <= mn);
	}
	return ok;
}

// ---------------------------------------------------------------------------------------
// Traversal state of one lane.  The stack is an LDS array indexed [entry][thread].
// Candidates compare by (t, rank): the reference keeps the first object of a leaf among
// equal t (src/scene.rs:105) and the right child among equal t (src/bvh.rs:98-104); `rank`
// encodes that order so the result does not depend on the visiting order.
// ---------------------------------------------------------------------------------------
// Where the per-lane stacks live.  A lane's stack pointer (Trav::sp) is kept in the unit its
// stack addresses with, so that a push or a pop is one add and one memory operation.
template <int kThreads> struct StackLds {  // LDS, [entry][thread]: conflict-free
	int* col;                              // this lane's column
	__device__ __forceinline__ int32_t bottom() const { return 0; }
	__device__ __forceinline__ void push(int32_t& sp, int32_t v) const { col[sp] = v; sp += kThreads; }
	__device__ __forceinline__ int32_t pop(int32_t& sp) const { sp -= kThreads; return col[sp]; }
};
// Global memory, [entry][every thread of the grid]: the lanes of a wave push to one 256-byte line,
// the array (tree depth x resident threads x 4 bytes: 16 MB for a depth of ten) stays in L2, and
// the vector-memory pipe that carries it is idle while the scene is read from LDS.  A lane reads
// back only what it wrote itself (program order), so no fence is involved.
struct StackMem {
	char* base;        // wave-uniform
	uint32_t first;    // byte offset of this lane's entry 0
	uint32_t stride;   // bytes from one entry to the next
	__device__ __forceinline__ int32_t bottom() const { return (int32_t)first; }
	__device__ __forceinline__ void push(int32_t& sp, int32_t v) const { *reinterpret_cast<int32_t*>(base + (uint32_t)sp) = v; sp += (int32_t)stride; }
	__device__ __forceinline__ int32_t pop(int32_t& sp) const { sp -= (int32_t)stride; return *reinterpret_cast<const int32_t*>(base + (uint32_t)sp); }
};

template <typename Real> struct Trav {
	int32_t cur;      // >= 0 inner node, < 0 leaf code, kDone when finished
	int32_t sp;       // in the stack's own unit, see StackLds / StackMem
	Real best_t;
	int32_t best_prim;  // index into prims, -1 = none
};
RT_FN void trav_begin(Trav<Real>& tv, int32_t stack_bottom) {
	tv.cur = 0; tv.sp = stack_bottom; tv.best_t = Fp<Real>::max_finite(); tv.best_prim = -1;
}
struct Tally { uint32_t box, sphere, cube; };
// Instrumented builds only: per code region, the lane-iterations executed (sum over active
// lanes) and the wave-iterations issued (counted by the first active lane), whose ratio / 64
// is the SIMD lane utilisation of that region.
enum : int { kRegInner = 0, kRegSphere, kRegCube, kRegLeaf, kRegShade, kRegSky, kRegLambert, kRegMetal, kRegGlass, kRegStart, kRegTravRound, kRegXPush, kRegXPop, kRegXFull, kRegXSpin, kRegPacket, kRegCount };
// Instrumented builds also clock the phases of the wave's main loop (s_memtime, wave-uniform).
enum : int { kPhShade = 0, kPhRegen, kPhFresh, kPhInner, kPhLeaf, kPhXchg, kPhPacket, kPhTotal, kPhCount };
struct Regions { uint32_t lane[kRegCount], wave[kRegCount]; unsigned long long clk[kPhCount]; };
template <bool kCount>
__device__ __forceinline__ unsigned long long phase_begin() {
	if constexpr (kCount) return __builtin_amdgcn_s_memtime();
	else return 0ull;
}
template <bool kCount>
__device__ __forceinline__ void phase_end(Regions& rg, int phase, unsigned long long t0) {
	if constexpr (kCount) rg.clk[phase] += __builtin_amdgcn_s_memtime() - t0;
}
// RT_CLOCK_ONLY (variant builds): instrumented launches keep the phase clocks but drop the per-iteration
// tallies, which slow the kernel 2.5x and distort the shares.
#ifdef RT_CLOCK_ONLY
constexpr bool kTally = false;
#else
constexpr bool kTally = true;
#endif
template <bool kCount>
__device__ __forceinline__ void region_tick(Regions& rg, int slot) {
	if constexpr (kCount && kTally) {
		rg.lane[slot]++;
		unsigned long long m = __ballot(1);
		if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) rg.wave[slot]++;
	}
}

template <typename Real, typename Stack>
__device__ __forceinline__ void trav_pop(Trav<Real>& tv, const Stack& stack) {
	if (tv.sp != stack.bottom()) {
		tv.cur = stack.pop(tv.sp);
	} else {
		tv.cur = kDone;
	}
}
// Descends inner nodes until the lane stands at a leaf or has run out of work, or until most
// lanes that entered with it have (it then stays at its inner node for the next round).
template <typename Real, typename Stack, bool kCount>
__device__ __forceinline__ void trav_inner(Trav<Real>& tv, const TravRay<Real>& r, const Node<Real>* nodes, const Stack& stack, Tally& tally, Regions& rg,
                                           int32_t inner_exit = 0) {
	const int32_t min_lanes = ((int32_t)__popcll(__ballot(tv.cur >= 0)) * inner_exit + 15) >> 4;  // fewer than inner_exit sixteenths of those that entered
	while (tv.cur >= 0) {
		if ((int32_t)__popcll(__ballot(1)) < min_lanes) break;  // the rest wait at leaves: serve them first
		region_tick<kCount>(rg, kRegInner);
		const Node<Real>& n = nodes[tv.cur];
		Real el, er;
		bool hl = hit_box<Real>(n.box, r, tv.best_t, &el);
		bool hr = hit_box<Real>(n.box + 6, r, tv.best_t, &er);
		int32_t cl = n.child[0], cr = n.child[1];
		if (kCount && kTally) tally.box += 2;
		if (hl && hr) {
			bool right_first = er < el;  // nearer child first, the left one on equal entry
			stack.push(tv.sp, right_first ? cl : cr);
			tv.cur = right_first ? cr : cl;
		} else if (hl) {
			tv.cur = cl;
		} else if (hr) {
			tv.cur = cr;
		} else {
			trav_pop<Real>(tv, stack);
		}
	}
}
// The same descent over quantised nodes (f32 trees in global memory): two 16-byte loads per
// visit instead of four, twelve conversions more.
template <typename Stack, bool kCount>
__device__ __forceinline__ void trav_inner_q(Trav<float>& tv, const TravRay<float>& r, const QNode* nodes, const Stack& stack, Tally& tally, Regions& rg,
                                             int32_t inner_exit) {
	const int32_t min_lanes = ((int32_t)__popcll(__ballot(tv.cur >= 0)) * inner_exit + 15) >> 4;
	while (tv.cur >= 0) {
		if ((int32_t)__popcll(__ballot(1)) < min_lanes) break;
		region_tick<kCount>(rg, kRegInner);
		const uint4* p = reinterpret_cast<const uint4*>(nodes + tv.cur);
		const uint4 a = p[0], b = p[1];
		// per child: three signed low planes, then three unsigned widths
		auto s_lo = [](uint32_t w) { return (float)(int16_t)(uint16_t)(w & 0xffffu); };
		auto s_hi = [](uint32_t w) { return (float)((int32_t)w >> 16); };
		auto u_lo = [](uint32_t w) { return (float)(w & 0xffffu); };
		auto u_hi = [](uint32_t w) { return (float)(w >> 16); };
		const float pl[12] = {s_lo(a.x), s_hi(a.x), s_lo(a.y), u_hi(a.y), u_lo(a.z), u_hi(a.z),
		                      s_lo(a.w), s_hi(a.w), s_lo(b.x), u_hi(b.x), u_lo(b.y), u_hi(b.y)};
		float el, er;
		bool hl = hit_box_q(pl, r.qscale, r.qneg, r.noi, r.eps, tv.best_t, &el);
		bool hr = hit_box_q(pl + 6, r.qscale, r.qneg, r.noi, r.eps, tv.best_t, &er);
		int32_t cl = (int32_t)b.z, cr = (int32_t)b.w;
		if (kCount && kTally) tally.box += 2;
		if (hl && hr) {
			bool right_first = er < el;
			stack.push(tv.sp, right_first ? cl : cr);
			tv.cur = right_first ? cr : cl;
		} else if (hl) {
			tv.cur = cl;
		} else if (hr) {
			tv.cur = cr;
		} else {
			trav_pop<float>(tv, stack);
		}
	}
}
// Tests the primitives of the leaf in tv.cur (the caller guarantees a leaf code) as one run of
// spheres and one run of cubes, then pops.
template <typename Real, bool kCount>
__device__ __forceinline__ void leaf_tests(Trav<Real>& tv, const TravRay<Real>& r, int32_t leaf, const Prim<Real>* prims, const Gate<Real>* gates, Tally& tally, Regions& rg) {
	region_tick<kCount>(rg, kRegLeaf);
	const uint32_t code = ~(uint32_t)leaf;
	const int32_t first = (int32_t)(code >> 6), ns = (int32_t)((code >> 3) & 7u), nc = (int32_t)(code & 7u);
	// Scene::hit accepts t > 0.001 && t < closest with closest starting at Scalar::MAX (src/scene.rs:102-105):
	// best_t starts at the largest finite value, so an infinite t never counts; among equal t the
	// larger rank wins, and there is nothing to tie with while no primitive has been accepted.
	auto better = [&](Real t, int32_t i) {
		return t > Real(0.001) && (t < tv.best_t || (t == tv.best_t && tv.best_prim >= 0 && prims[i].rank > prims[tv.best_prim].rank));
	};
	for (int32_t k = 0; k < ns; k++) {
		int32_t i = first + k;
		Real t;
		region_tick<kCount>(rg, kRegSphere);
		if (kCount && kTally) tally.sphere++;
		if (hit_sphere<Real>(prims[i], r, &t) && better(t, i)) {
			// a gated (inverted) sphere counts only if the reference would have reached it
			const uint32_t n_gates = prim_gate_count(prims[i]), first_gate = prim_gate_first(prims[i]);
			bool reachable = true;
			for (uint32_t g = 0; g < n_gates; g++) reachable = reachable && hit_box_reference<Real>(gates[first_gate + g].box, r);
			if (reachable) { tv.best_t = t; tv.best_prim = i; }
		}
	}
	for (int32_t k = 0; k < nc; k++) {
		int32_t i = first + ns + k;
		Real t;
		region_tick<kCount>(rg, kRegCube);
		if (kCount && kTally) tally.cube++;
		if (hit_cube<Real>(prims[i], r, &t) && better(t, i)) { tv.best_t = t; tv.best_prim = i; }
	}
}
template <typename Real, typename Stack, bool kCount>
__device__ __forceinline__ void trav_leaf(Trav<Real>& tv, const TravRay<Real>& r, const Prim<Real>* prims, const Gate<Real>* gates, const Stack& stack, Tally& tally, Regions& rg) {
	leaf_tests<Real, kCount>(tv, r, tv.cur, prims, gates, tally, rg);
	trav_pop<Real>(tv, stack);
}

// The 64 camera rays of one batch (an 8x8 tile, one sample) walked through the tree TOGETHER: one
// wave-uniform node index, one stack for the wave (a register, entry k in lane k), every lane
// tests its own ray against the two child boxes and the wave descends wherever any lane has to go,
// nearer child first by majority.  Such rays differ by a pixel and a lens offset, so the union of
// their walks is hardly longer than one walk, the node index is uniform (an LDS broadcast, scalar
// control flow, no per-lane stack traffic) and all 64 lanes work in every step, where rays of
// their own fill 40-70 % of the lanes in the same loops.  The closest hit of a ray does not depend
// on the order its candidates are found in (ties go by rank), and the primitive tests decide
// alone what counts, so a lane may test primitives its own walk would have culled: the result is
// word for word the per-ray walk's.
// A record read word by word through the pointer's own address space: with a pointer into the
// constant address space and a wave-uniform index these are scalar loads.
typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
template <typename T, typename Words>
__device__ __forceinline__ T load_record(Words p) {
	static_assert(sizeof(T) % 4 == 0, "records are made of 32-bit words");
	uint32_t w[sizeof(T) / 4];
#pragma unroll
	for (uint32_t i = 0; i < sizeof(T) / 4; i++) w[i] = p[i];
	T out;
	__builtin_memcpy(&out, w, sizeof(T));
	return out;
}
// leaf_tests for a packet: the leaf is the wave's, so every record is read once for the wave.
template <typename Real, bool kCount, typename Words>
__device__ __forceinline__ void packet_leaf_tests(Trav<Real>& tv, const TravRay<Real>& r, int32_t leaf, Words prims, Words gates, Tally& tally, Regions& rg) {
	constexpr uint32_t kPrimWords = sizeof(Prim<Real>) / 4, kGateWords = sizeof(Gate<Real>) / 4, kRankWord = offsetof(Prim<Real>, rank) / 4;
	region_tick<kCount>(rg, kRegLeaf);
	const uint32_t code = ~(uint32_t)leaf;
	const int32_t first = (int32_t)(code >> 6), ns = (int32_t)((code >> 3) & 7u), nc = (int32_t)(code & 7u);
	auto better = [&](Real t, int32_t rank) {  // see leaf_tests
		return t > Real(0.001) && (t < tv.best_t || (t == tv.best_t && tv.best_prim >= 0 && rank > (int32_t)prims[(uint32_t)tv.best_prim * kPrimWords + kRankWord]));
	};
	for (int32_t k = 0; k < ns; k++) {
		const int32_t i = first + k;
		const Prim<Real> pr = load_record<Prim<Real>>(prims + (uint32_t)i * kPrimWords);
		Real t;
		region_tick<kCount>(rg, kRegSphere);
		if (kCount && kTally) tally.sphere++;
		if (hit_sphere<Real>(pr, r, &t) && better(t, pr.rank)) {
			const uint32_t n_gates = prim_gate_count(pr), first_gate = prim_gate_first(pr);
			bool reachable = true;
			for (uint32_t g = 0; g < n_gates; g++) {
				const Gate<Real> gate = load_record<Gate<Real>>(gates + (first_gate + g) * kGateWords);
				reachable = reachable && hit_box_reference<Real>(gate.box, r);
			}
			if (reachable) { tv.best_t = t; tv.best_prim = i; }
		}
	}
	for (int32_t k = 0; k < nc; k++) {
		const int32_t i = first + ns + k;
		const Prim<Real> pr = load_record<Prim<Real>>(prims + (uint32_t)i * kPrimWords);
		Real t;
		region_tick<kCount>(rg, kRegCube);
		if (kCount && kTally) tally.cube++;
		if (hit_cube<Real>(pr, r, &t) && better(t, pr.rank)) { tv.best_t = t; tv.best_prim = i; }
	}
}
template <typename Real, bool kCount, typename Words>
__device__ __forceinline__ void packet_trace(Trav<Real>& pt, const TravRay<Real>& r, bool valid, Words nodes, Words prims, Words gates,
                                             int32_t big_leaf, int32_t start, Tally& tally, Regions& rg) {
	constexpr uint32_t kNodeWords = sizeof(Node<Real>) / 4;
	const int32_t lane = (int32_t)(threadIdx.x & 63u);
	trav_begin<Real>(pt, 0);
	int32_t cur = 0;
	if (big_leaf != 0) {
		packet_leaf_tests<Real, kCount, Words>(pt, r, big_leaf, prims, gates, tally, rg);
		cur = start;
	}
	if (start == kDone) cur = kDone;
	int32_t sp = 0;
	int wstack = 0;  // the wave's stack: entry k lives in lane k (tree depth <= 64, checked by the host)
	for (;;) {
		while (cur >= 0) {
			region_tick<kCount>(rg, kRegPacket);
			const Node<Real> n = load_record<Node<Real>>(nodes + (uint32_t)cur * kNodeWords);
			Real el, er;
			const bool hl = hit_box<Real>(n.box, r, pt.best_t, &el) && valid;
			const bool hr = hit_box<Real>(n.box + 6, r, pt.best_t, &er) && valid;
			const int32_t cl = __builtin_amdgcn_readfirstlane(n.child[0]), cr = __builtin_amdgcn_readfirstlane(n.child[1]);
			if (kCount && kTally) tally.box += 2;
			const unsigned long long ml = __ballot(hl), mr = __ballot(hr);
			if (ml != 0ull && mr != 0ull) {
				const int32_t right_votes = (int32_t)__popcll(__ballot(hr && (!hl || er < el)));
				const int32_t left_votes = (int32_t)__popcll(__ballot(hl && (!hr || !(er < el))));
				const bool right_first = right_votes > left_votes;
				wstack = lane == sp ? (right_first ? cl : cr) : wstack;
				sp++;
				cur = right_first ? cr : cl;
			} else if (ml != 0ull) {
				cur = cl;
			} else if (mr != 0ull) {
				cur = cr;
			} else if (sp != 0) {
				sp--;
				cur = __builtin_amdgcn_readlane(wstack, sp);
			} else {
				cur = kDone;
			}
		}
		if (cur == kDone) break;
		packet_leaf_tests<Real, kCount, Words>(pt, r, cur, prims, gates, tally, rg);
		if (sp != 0) {
			sp--;
			cur = __builtin_amdgcn_readlane(wstack, sp);
		} else {
			cur = kDone;
		}
	}
}

// BVH::trace, src/bvh.rs:130-142, with get_normal of src/shapes/sphere.rs:60-62 and
// src/shapes/cube.rs:54-64 (Rust signum: +1 for +0.0, -1 for -0.0).
template <typename Real> struct Hit {
	V3<Real> p, n;
	bool front;
	int32_t id;        // object id
	int32_t material;  // RTHIP_LAMBERTIAN | RTHIP_METAL | RTHIP_DIELECTRIC
};
RT_FN Hit<Real> make_hit(const Prim<Real>& pr, bool is_sphere, const TravRay<Real>& r, Real t) {
	Hit<Real> h;
	h.p = r.o + r.d * t;
	if (is_sphere) {
		if constexpr (std::is_same<Real, float>::value) h.n = (h.p - ld3(pr.v)) * Fp<float>::rcp(pr.v[3]);
		else h.n = (h.p - ld3(pr.v)) / pr.v[3];
	} else {
		V3<Real> centre;
		if constexpr (std::is_same<Real, float>::value)
			centre = mk<Real>(pr.v[kBoxLoX] + pr.v[kBoxHiX], pr.v[kBoxLoY] + pr.v[kBoxHiY], pr.v[kBoxLoZ] + pr.v[kBoxHiZ]) * 0.5f;
		else centre = ld3(pr.v);
		V3<Real> rel = h.p - centre;
		Real ax = Fp<Real>::fabs(rel.x), ay = Fp<Real>::fabs(rel.y), az = Fp<Real>::fabs(rel.z);
		Real maxc = Fp<Real>::fmax(Fp<Real>::fmax(ax, ay), az);
		if (maxc == ax) h.n = mk<Real>(Fp<Real>::copysign(Real(1), rel.x), Real(0), Real(0));
		else if (maxc == ay) h.n = mk<Real>(Real(0), Fp<Real>::copysign(Real(1), rel.y), Real(0));
		else h.n = mk<Real>(Real(0), Real(0), Fp<Real>::copysign(Real(1), rel.z));
	}
	h.front = dot(r.d, h.n) < Real(0);
	if (!h.front) h.n = -h.n;
	h.id = pr.id & ((1 << kMatShift) - 1);
	h.material = (pr.id >> kMatShift) & 3;
	return h;
}

// Material::reflectance, src/materials.rs:80-85 (Schlick).  One departure from the reference in BOTH
// precisions: (1 - cos)^5 is a multiply chain here where the reference calls powf(5.); the two agree
// to a few ulps, not bit for bit, which is why per-sample radiance is compared bit-exactly only with
// a CPU run that uses the same chain, and to 1e-12 relative with one that calls pow().
RT_FN Real reflectance(Real cosine, Real ref_idx) {
	Real r0;
	if constexpr (std::is_same<Real, float>::value) r0 = (1.0f - ref_idx) * Fp<float>::rcp(1.0f + ref_idx);
	else r0 = (Real(1) - ref_idx) / (Real(1) + ref_idx);
	r0 = r0 * r0;
	Real x = Real(1) - cosine;
	Real x2 = x * x;
	return r0 + (Real(1) - r0) * ((x2 * x2) * x);
}
template <typename Real> struct Scattered {
	V3<Real> att, dir;
	bool ok;
};
// Material::scatter, src/materials.rs:17-68.  ok = false for None (an absorbed metal ray).
// Draw counts: Lambertian 3, Metal 3 (also when fuzz is 0), Dielectric 1 unless total
// internal reflection (the `||` of :54-55 short-circuits).
RT_FN Scattered<Real> scatter(const Hit<Real>& h, const Mat<Real>& m, V3<Real> unit_dir, Rng& rng) {  // unit_dir = normalize(ray direction)
	Scattered<Real> out;
	out.ok = true;
	if (h.material != RTHIP_DIELECTRIC) {
		// Lambertian: n + ruv (near-zero -> n).  Metal: reflect(unit_dir, n) + fuzz * ruv, absorbed when it
		// points into the surface.  Both draw one random unit vector, so the lanes of either kind share one
		// instruction stream: base + scale * ruv with (n, 1) or (reflected, fuzz); 1 * ruv is exact.
		const bool metal = h.material == RTHIP_METAL;
		const V3<Real> ruv = random_unit_vector<Real>(rng);
		const V3<Real> reflected = reflect(unit_dir, h.n);
		const V3<Real> base = metal ? reflected : h.n;
		const Real scale = metal ? m.param : Real(1);
		V3<Real> sd = base + scale * ruv;
		const Real tol = Real(0.00000001);
		if (!metal && Fp<Real>::fabs(sd.x) <= tol && Fp<Real>::fabs(sd.y) <= tol && Fp<Real>::fabs(sd.z) <= tol) sd = h.n;
		out.ok = !(metal && dot(sd, h.n) <= Real(0));
		out.att = ld3(m.albedo);
		out.dir = sd;
	} else {
		Real ir = m.param;
		Real ratio = h.front ? Fp<Real>::rcp(ir) : ir;
		Real cos_theta = Fp<Real>::fmin(dot(-unit_dir, h.n), Real(1));
		Real sin_theta = Fp<Real>::sqrt(Real(1) - (cos_theta * cos_theta));
		bool cannot_refract = ratio * sin_theta > Real(1);
		if (cannot_refract || (reflectance<Real>(cos_theta, ratio) > random_range<Real>(Real(0), Real(1), rng))) {
			out.dir = reflect(unit_dir, h.n);
		} else {
			out.dir = refract(unit_dir, h.n, ratio);
		}
		out.att = mk<Real>(Real(1), Real(1), Real(1));
	}
	return out;
}

// sqrt(radiance) -> fixed point (2^-36 units).  NaN and negatives count as 0; one sample is
// capped at 256 (radiance 65536), far above anything a display shows, so sums cannot wrap
// below 2^20 samples per pixel.  f32: the value has 24 significant bits, so splitting it into
// an integer part in 2^-4 units and a 32-bit fraction of that is exact and needs two
// hardware conversions instead of an emulated 64-bit one.
RT_FN unsigned long long to_fixed(Real v) {
	if constexpr (std::is_same<Real, float>::value) {
		float x = v * 16.0f;                        // units of 2^-4; NaN stays NaN
		x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 4096.0f);  // NaN -> 0, cap at 256
		float hi = __builtin_truncf(x);
		float lo = (x - hi) * 4294967296.0f;        // fraction in units of 2^-36, < 2^32, exact
		return ((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo;
	} else {
		double x = (double)v;
		if (!(x > 0.0)) return 0ull;
		if (x > 256.0) x = 256.0;
		return (unsigned long long)(x * kAccumScale + 0.5);
	}
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// One bounce of ray_color (src/renderer.rs:14-25) in its unrolled form:
//   miss     -> L += T * (sky * T), the path ends      (:24; the sky term is unwound through the attenuations again)
//   absorbed -> the path ends, this hit's emission is dropped (:20-22)
//   else     -> L += T * emission, T *= attenuation, depth - 1; depth 0 ends the path (:10-12)
// Returns true when the path continues along *next.
RT_FN bool shade_ray(const TravRay<Real>& ray, Real best_t, int32_t best_prim, const Prim<Real>* prims, const Mat<Real>* mats,
                     int sky_mode, Rng& rng, V3<Real>& T, V3<Real>& L, int32_t& depth, RayT<Real>* next) {
	const V3<Real> unit_dir = normalize(ray.d);  // sky, metal and glass all start from it: once per wave
	if (best_prim < 0) {
		L = L + T * (sky<Real>(unit_dir, sky_mode) * T);
		return false;
	}
	const Prim<Real>& pr = prims[best_prim];
	// the material record comes from global memory: ask for it before the hit record is worked out
	const Mat<Real> mat = mats[best_prim];  // materials are stored in primitive order: no trip through the primitive's id
	Hit<Real> h = make_hit<Real>(pr, ((pr.id >> kKindShift) & 1) == RTHIP_SPHERE, ray, best_t);
	Scattered<Real> sc = scatter<Real>(h, mat, unit_dir, rng);
	if (!sc.ok) return false;
	if (h.material == RTHIP_LAMBERTIAN) L = L + T * (ld3(mat.albedo) * mat.param);  // materials.rs:71-77
	T = T * sc.att;
	depth--;
	if (depth <= 0) return false;
	next->o = h.p;
	next->d = sc.dir;
	return true;
}
// raytracer_headless/src/main.rs:53: gamma (sqrt) per sample, then summed.  The buffer is
// [tile][channel][64 px]: the lanes of a wave that finish together touch few cache lines per
// atomic instruction.
RT_FN void accumulate_sample(unsigned long long* accum, uint32_t accum_index, V3<Real> L) {
	unsigned long long* acc = accum + (size_t)(accum_index >> 6) * (kTilePixels * kAccumChannels) + (accum_index & 63u);
	atomicAdd(acc, to_fixed<Real>(Fp<Real>::sqrt(L.x)));
	atomicAdd(acc + kTilePixels, to_fixed<Real>(Fp<Real>::sqrt(L.y)));
	atomicAdd(acc + 2 * kTilePixels, to_fixed<Real>(Fp<Real>::sqrt(L.z)));
}
// Work-item decoding shared by every variant: path j of work item `item` (an 8x8 tile x `chunk`
// samples).  Returns false for pixels outside the image and samples beyond spp_count.
struct PathId { uint32_t px, py, accum_index, s_local; };
__device__ __forceinline__ bool decode_path(uint32_t item, uint32_t j, uint32_t n_chunks, uint32_t chunk, uint32_t spp_count,
                                            int32_t part_index, int32_t part_count, int32_t tiles_x, int32_t width, int32_t height, PathId* out, uint32_t block_order = 0u,
                                            bool sample_major = false) {
	uint32_t tile_local = item / n_chunks, chunk_i = item - tile_local * n_chunks;
	uint32_t tile = (uint32_t)part_index + tile_local * (uint32_t)part_count;
	uint32_t ty = tile / (uint32_t)tiles_x, tx = tile - ty * (uint32_t)tiles_x;
	if (block_order != 0u && part_count == 1) {
		// Work order in blocks of block_order x block_order tiles (32 by default; row-major inside a block, blocks row-major): the
		// waves at work at one time then look at a compact part of the scene, which is what the
		// L2 of an XCD can hold of a tree that lives in memory.  The tile's place in the
		// accumulation buffer does not change.
		const uint32_t S = block_order, tiles_y = ((uint32_t)height + kTile - 1u) / kTile;
		const uint32_t band = tile_local / ((uint32_t)tiles_x * S), r = tile_local - band * (uint32_t)tiles_x * S;
		const uint32_t h = min(S, tiles_y - band * S);
		const uint32_t n_cols = ((uint32_t)tiles_x + S - 1u) / S;
		const uint32_t sx = min(r / (S * h), n_cols - 1u), rr = r - sx * S * h;
		const uint32_t w = min(S, (uint32_t)tiles_x - sx * S);
		const uint32_t ly = rr / w, lx = rr - ly * w;
		tx = sx * S + lx; ty = band * S + ly;
		tile = ty * (uint32_t)tiles_x + tx;
	}
	// Which path of the item path j is.  By default the item is run sample by sample: 64 consecutive paths are
	// one sample of the tile's 64 pixels.  sample_major runs it pixel by pixel, the pixels in Morton order:
	// 64 consecutive paths are then all samples of a few neighbouring pixels, i.e. rays that differ by a
	// pixel or two, which is what a packet of camera rays wants (the union of its rays' walks stays short).
	uint32_t x = j & 7u, y = (j >> 3) & 7u, s_in = j >> 6;
	if (sample_major) {
		uint32_t q = (uint32_t)((float)j * __builtin_amdgcn_rcpf((float)chunk));  // j / chunk (j < 64 * chunk): the hardware reciprocal is an ulp off at most,
		if (q * chunk > j) q--;                                                   // so one step either way settles it
		else if ((q + 1u) * chunk <= j) q++;
		s_in = j - q * chunk;
		x = (q & 1u) | ((q >> 1) & 2u) | ((q >> 2) & 4u);
		y = ((q >> 1) & 1u) | ((q >> 2) & 2u) | ((q >> 3) & 4u);
	}
	out->px = tx * kTile + x;
	out->py = ty * kTile + y;
	out->s_local = chunk_i * chunk + s_in;
	out->accum_index = tile * kTilePixels + (y * 8u + x);
	return out->px < (uint32_t)width && out->py < (uint32_t)height && out->s_local < spp_count;
}

enum : int { kIdle = 0, kTraverse = 1, kShade = 2, kFresh = 3 };

// ---------------------------------------------------------------------------------------
// Exchange queues (kXchg): rays that wait to be shaded change lanes, and waves, so that one
// shading step serves one kind of ray.  The shading code has three arms that a wave runs one
// after the other, each on the lanes that take it: a miss (sky, sample to the accumulation
// buffer), a Lambertian or metal hit, a glass hit; a wave that leaves its traversal loop holds
// a mix of all three.  With the queues, the lanes of the minority arms put their rays into the
// workgroup's queue for that arm (80 bytes of LDS per ray) and free lanes take rays of the
// majority arm out of its queue, so the step runs one arm on nearly every lane.  A queue that
// is half full is served first whatever the wave holds.  Nothing waits for a queue: a ray that
// finds its queue full is shaded where it is, as without the queues, and a wave only takes
// what is there.  Which lane shades a ray does not change the ray, so images stay bit-identical.
//
// A queue is a ring of `cap` records claimed by tickets: `tail` hands out positions to writers
// and `head` to readers, `space` and `avail` admit only as many of either as there are records
// free or written, and every record carries the number of the position it is ready for
// (seq == position: free for the writer of that position; position + 1: written, for its
// reader; the reader leaves position + cap).  Waits are therefore only ever for a wave that is
// in the middle of copying a record, never for one that has yet to decide something.
// ---------------------------------------------------------------------------------------
constexpr int kArms = 3;
constexpr uint32_t kXchgRecBytes = 80;
__device__ __forceinline__ uint32_t xchg_peek(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int32_t xchg_peek(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
struct Xchg {
	char* base;      // control words, one 16-byte row per counter and a column per arm: avail, space, head, tail; then seq, then the records
	uint32_t cap;
	__device__ __forceinline__ int32_t* avail() const { return reinterpret_cast<int32_t*>(base); }
	__device__ __forceinline__ int32_t* space() const { return reinterpret_cast<int32_t*>(base + 16); }
	__device__ __forceinline__ uint32_t* head() const { return reinterpret_cast<uint32_t*>(base + 32); }
	__device__ __forceinline__ uint32_t* tail() const { return reinterpret_cast<uint32_t*>(base + 48); }
	// one counter of all arms at a glance (wave-uniform)
	__device__ __forceinline__ int4 peek4(const int32_t* row) const {
		return make_int4(__builtin_amdgcn_readfirstlane(xchg_peek(row)), __builtin_amdgcn_readfirstlane(xchg_peek(row + 1)), __builtin_amdgcn_readfirstlane(xchg_peek(row + 2)), 0);
	}
	__device__ __forceinline__ uint32_t* seq(int arm) const { return reinterpret_cast<uint32_t*>(base + 64) + (uint32_t)arm * cap; }
	__device__ __forceinline__ uint4* rec(int arm, uint32_t slot) const { return reinterpret_cast<uint4*>(base + 64 + kArms * cap * 4u + ((uint32_t)arm * cap + slot) * kXchgRecBytes); }
};
__host__ __device__ constexpr uint32_t xchg_bytes(uint32_t cap) { return 64u + kArms * cap * (4u + kXchgRecBytes); }
// LDS operations of one wave are carried out in the order they were issued: a compiler barrier
// between the record and its sequence word is all the ordering a writer needs, and likewise
// between a reader's look at the sequence word and its loads of the record.
__device__ __forceinline__ void xchg_order() { asm volatile("" ::: "memory"); }

// ---------------------------------------------------------------------------------------
// The megakernel.
// ---------------------------------------------------------------------------------------
// kLds: 0 = the scene is read from global memory; 1 = nodes, primitives and gates are copied to LDS;
// 2 = the materials as well (they are read once per ray, and that one trip to L2 per bounce is worth
// 6-7 % when LDS has the room).
// kAccLds: finished paths add their sample to a 192-word tile that the wave keeps in LDS, and the
// tile goes to the accumulation buffer once per work item (one atomic per word per item instead of
// three per path).  A wave then finishes every path of an item before it takes the next one, so that
// its tile belongs to one item at a time.  For scenes whose paths are a ray or two long, where the
// accumulation atomics were what bounded the kernel (the three-object 4K scene ran at the same speed
// with and without its camera's defocus arithmetic), and only where LDS has the room.
// kStackMem: the lanes' traversal stacks in global memory (StackMem) instead of LDS.
// kXchg: rays waiting to be shaded are regrouped by shading arm through queues in LDS (see Xchg).
// kRootOnly: the scene is one leaf (RenderArgs::start == kDone): every ray tests that leaf when it is set up and goes
// on to shading; without the traversal code the kernel needs 70 registers and seven waves per SIMD fit.
template <typename Real, int kThreads, int kLds, bool kCount, bool kQ = false, bool kAccLds = false, bool kStackMem = false, bool kXchg = false, bool kRootOnly = false>
__global__ __launch_bounds__(kThreads, (std::is_same<Real, float>::value && RT_MINWAVES ? (kThreads == 1024 ? 8 : (kRootOnly ? RT_ROOTWAVES : (kQ ? RT_QWAVES : RT_MINWAVES))) : 2)) void render_kernel(RenderArgs<Real> A) {
	static_assert(!kRootOnly || (kAccLds && !kQ && !kStackMem && !kXchg), "the one-leaf kernel exists with accumulation tiles only");
	static_assert(!kQ || (kLds == 0 && std::is_same<Real, float>::value), "quantised nodes: f32 trees in global memory only");
	static_assert(!kXchg || (kLds != 0 && !kAccLds && std::is_same<Real, float>::value), "exchange queues: f32, scene in LDS, no accumulation tiles");
	extern __shared__ __align__(64) char smem[];
	using Stack = typename std::conditional<kStackMem, StackMem, StackLds<kThreads>>::type;
	Stack stack;
	uint32_t stack_bytes = 0;
	if constexpr (kStackMem) {
		stack.base = A.stack_mem;
		stack.first = (blockIdx.x * (uint32_t)kThreads + threadIdx.x) * 4u;
		stack.stride = gridDim.x * (uint32_t)kThreads * 4u;
	} else {
		stack.col = reinterpret_cast<int*>(smem) + threadIdx.x;  // [entry * kThreads]
		stack_bytes = (uint32_t)A.stack_entries * kThreads * 4u;
	}
	const Node<Real>* nodes;
	const Prim<Real>* prims;
	const Mat<Real>* mats;
	const Gate<Real>* gates;
	Xchg xq{nullptr, 0u};
	if constexpr (kXchg) {
		xq.base = smem + A.xchg_off;
		xq.cap = A.xchg_cap;
		for (uint32_t i = threadIdx.x; i < kArms * xq.cap; i += kThreads) xq.seq(0)[i] = i & (xq.cap - 1u);
		if (threadIdx.x < 16u) reinterpret_cast<uint32_t*>(xq.base)[threadIdx.x] = (threadIdx.x >> 2) == 1u ? xq.cap : 0u;
	}
	if constexpr (kLds) {
		char* dst = smem + stack_bytes;
		const uint4* src = reinterpret_cast<const uint4*>(A.scene);
		for (uint32_t i = threadIdx.x; i < A.lds_bytes / 16u; i += kThreads) reinterpret_cast<uint4*>(dst)[i] = src[i];
		__syncthreads();
		nodes = reinterpret_cast<const Node<Real>*>(dst);
		prims = reinterpret_cast<const Prim<Real>*>(dst + A.prims_off);
		if constexpr (kLds == 2) mats = reinterpret_cast<const Mat<Real>*>(dst + A.mats_off);
		else mats = reinterpret_cast<const Mat<Real>*>(A.scene + A.mats_off);  // read once per ray: from L2 when LDS has no room
		gates = reinterpret_cast<const Gate<Real>*>(dst + A.gates_off);
	} else {
		nodes = reinterpret_cast<const Node<Real>*>(A.scene);
		prims = reinterpret_cast<const Prim<Real>*>(A.scene + A.prims_off);
		mats = reinterpret_cast<const Mat<Real>*>(A.scene + A.mats_off);
		gates = reinterpret_cast<const Gate<Real>*>(A.scene + A.gates_off);
	}
	const uint32_t lane = threadIdx.x & 63u;
	// Per-sample radiance (samples_out: parity tests and probes) is written by the f64 kernels and by the
	// instrumented f32 ones; the f32 production kernels do not carry a path's sample slot along.
	constexpr bool kSamples = kCount || !std::is_same<Real, float>::value;
	// the wave's accumulation tile (kAccLds): [channel][64 px] words behind the stacks and the scene copy
	unsigned long long* acc_tile = nullptr;
	unsigned long long* acc_dst = A.accum;  // where the tile belongs in the accumulation buffer
	if constexpr (kAccLds) {
		acc_tile = reinterpret_cast<unsigned long long*>(smem + A.acc_tile_off) + (threadIdx.x >> 6) * (kTilePixels * kAccumChannels);
		for (int k = 0; k < kAccumChannels; k++) acc_tile[k * kTilePixels + lane] = 0ull;
	}
	auto flush_tile = [&]() {  // every lane of the wave: its pixel's three words
		if constexpr (kAccLds) {
			for (int k = 0; k < kAccumChannels; k++) {
				const unsigned long long v = acc_tile[k * kTilePixels + lane];
				if (v) { atomicAdd(acc_dst + k * kTilePixels + lane, v); acc_tile[k * kTilePixels + lane] = 0ull; }
			}
		}
	};

	// lane state
	int state = kIdle;
	TravRay<Real> ray;
	Trav<Real> tv;
	V3<Real> T, L;
	int32_t depth = 0;
	Rng rng;
	uint32_t accum_index = 0, sample_slot = 0;
	Tally tally{0, 0, 0};
	Regions rg{};
	uint32_t n_paths = 0;                // per lane: kernels that start their paths on demand
	uint32_t w_rays = 0, w_paths = 0;    // per wave (scalar registers): rays shaded, paths started in batches
	tv.cur = kDone; tv.sp = 0; tv.best_t = 0; tv.best_prim = -1;
	rng.state = 0; rng.inc = 1;
	T = L = mk<Real>(0, 0, 0);
	ray.o = ray.d = ray.inv = ray.noi = ray.qscale = ray.qneg = T; ray.eps = ray.rcp_a = 0;

	// wave state (uniform)
	// Work items are cut into kQueues contiguous ranges (bands of the image), each with its own
	// counter.  Workgroups whose index agrees modulo 8 share an XCD and its L2, so they start on the
	// same range and the rays an L2 serves stay in one part of the scene; a wave whose range is
	// exhausted moves on to the next one (all ranges get finished whatever the placement).
	const unsigned long long t_kernel = phase_begin<kCount>();
	uint32_t item = 0, pool_pos = 0, pool_end = 0, queue_hop = 0;
	const uint32_t home_queue = blockIdx.x % kQueues;
	bool queue_dry = false;
	const uint32_t pool_size = 64u * A.chunk;
	// f32 kernels make their camera rays in batches (see the regeneration step)
#ifdef RT_NO_BATCH_REGEN
	constexpr bool kBatchRegen = false;
#else
	constexpr bool kBatchRegen = std::is_same<Real, float>::value;
#endif
	V3<Real> sp_o = mk<Real>(0, 0, 0), sp_d = sp_o;
	uint32_t sp_state = 0, sp_inc = 1, sp_accum = 0xFFFFFFFFu, sp_slot = 0;
	uint32_t spare_pos = 0, spare_end = 0;
	// Kernels of scenes in global memory are short of registers: their spares keep origin, direction and hit
	// only, and the lane that takes one works out pixel, slots and random stream again from the batch's
	// place in its work item (the stream after the camera's draws, which are the same number for every ray).
	constexpr bool kLeanSpares = kLds == 0;
	uint32_t sp_item = 0, sp_base = 0;
	// ... and, where the launch asks for it, walks them through the tree as one packet (see packet_trace):
	// the spares then carry their hit, and the lanes that take them start with shading.
#ifdef RT_PACKET_LDS
	constexpr bool kPacket = kBatchRegen;
#else
	constexpr bool kPacket = kBatchRegen && kLds == 0;  // scenes in LDS: measured slower (see packet_trace; the three-object scene as well: 83 against 87 Gpaths/s), and the code costs registers
#endif
	uint32_t sp_fh_state = 0, sp_fh_hit = 0;

	for (;;) {
		unsigned long long t_ph = phase_begin<kCount>();
		// With the exchange queues, shading steps follow one another for as long as there are free
		// lanes and a queue that can fill them: each step runs one arm.
		for (;;) {
		// ---- exchange: one shading arm per step ---------------------------------------------
		if constexpr (kXchg) {
			int arm = -1;
			if (state == kShade) arm = tv.best_prim < 0 ? 0 : ((((prims[tv.best_prim].id >> kMatShift) & 3) == RTHIP_DIELECTRIC) ? 2 : 1);
			const unsigned long long m_arm[kArms] = {__ballot(arm == 0), __ballot(arm == 1), __ballot(arm == 2)};
			const int32_t n_arm[kArms] = {(int32_t)__popcll(m_arm[0]), (int32_t)__popcll(m_arm[1]), (int32_t)__popcll(m_arm[2])};
			const int32_t n_idle = (int32_t)__popcll(__ballot(state == kIdle));
			const int4 av4 = xq.peek4(xq.avail()), sp4 = xq.peek4(xq.space());
			const int32_t avail[kArms] = {max(0, av4.x), max(0, av4.y), max(0, av4.z)};
			const int32_t space[kArms] = {sp4.x, sp4.y, sp4.z};
			// the arm this step runs: the one that fills most lanes, counting what free lanes could take
			// out of its queue and what cannot leave because its own queue is full; a queue at its
			// half-way mark goes first
			int best = 0, best_score = -1;
			for (int k = 0; k < kArms; k++) {
				int32_t room = n_idle;
				for (int j = 0; j < kArms; j++) if (j != k && space[j] >= n_arm[j] && n_arm[j] <= A.xchg_push_max) room += n_arm[j];
				const int32_t score = n_arm[k] + min(avail[k], room) + (avail[k] >= (int32_t)(xq.cap / 2u) ? 64 : 0);
				if (score > best_score) { best_score = score; best = k; }
			}
			// the other arms' lanes hand their rays over (a full queue: they are shaded where they are)
			for (int j = 0; j < kArms; j++) {
				if (j == best || n_arm[j] == 0 || n_arm[j] > A.xchg_push_max) continue;
				if (space[j] < n_arm[j]) {
					if (arm == j) region_tick<kCount>(rg, kRegXFull);
					continue;
				}
				int32_t had = 0;
				uint32_t pos = 0;
				if (lane == 0) {
					had = atomicSub(xq.space() + j, n_arm[j]);
					if (had >= n_arm[j]) pos = atomicAdd(xq.tail() + j, (uint32_t)n_arm[j]);
					else atomicAdd(xq.space() + j, n_arm[j]);
				}
				had = __builtin_amdgcn_readfirstlane(had);
				pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
				if (had < n_arm[j]) {
					if (arm == j) region_tick<kCount>(rg, kRegXFull);
					continue;
				}
				if (arm == j) {
					region_tick<kCount>(rg, kRegXPush);
					const uint32_t p = pos + lane_rank(m_arm[j]);
					const uint32_t slot = p & (xq.cap - 1u);
					uint32_t* sq = xq.seq(j) + slot;
					while (xchg_peek(sq) != p) { region_tick<kCount>(rg, kRegXSpin); __builtin_amdgcn_s_sleep(1); }
					xchg_order();
					uint4* r = xq.rec(j, slot);
					r[0] = make_uint4(__float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z), __float_as_uint(ray.d.x));
					r[1] = make_uint4(__float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(tv.best_t), (uint32_t)tv.best_prim);
					r[2] = make_uint4(__float_as_uint(T.x), __float_as_uint(T.y), __float_as_uint(T.z), __float_as_uint(L.x));
					r[3] = make_uint4(__float_as_uint(L.y), __float_as_uint(L.z), rng.state, rng.inc);
					r[4] = make_uint4(accum_index, (uint32_t)depth, sample_slot, 0u);
					xchg_order();
					__hip_atomic_store(sq, p + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					state = kIdle;
				}
				xchg_order();
				if (lane == 0) atomicAdd(xq.avail() + j, n_arm[j]);
			}
			// free lanes take rays of the chosen arm: enough of them to be worth the step, or whatever
			// there is once the wave has no other work
			const unsigned long long m_free = __ballot(state == kIdle);
			const int32_t want = min((int32_t)__popcll(m_free), avail[best]);
			const bool starving = queue_dry && pool_pos == pool_end && spare_pos == spare_end;
			if (want > 0 && (want >= A.xchg_pop_min || starving)) {
				int32_t had = 0;
				uint32_t pos = 0;
				if (lane == 0) {
					had = atomicSub(xq.avail() + best, want);
					if (had >= want) pos = atomicAdd(xq.head() + best, (uint32_t)want);
					else atomicAdd(xq.avail() + best, want);
				}
				had = __builtin_amdgcn_readfirstlane(had);
				pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
				if (had >= want) {
					const uint32_t rk = lane_rank(m_free);
					if (state == kIdle && rk < (uint32_t)want) {
						region_tick<kCount>(rg, kRegXPop);
						const uint32_t p = pos + rk;
						const uint32_t slot = p & (xq.cap - 1u);
						uint32_t* sq = xq.seq(best) + slot;
						while (xchg_peek(sq) != p + 1u) { region_tick<kCount>(rg, kRegXSpin); __builtin_amdgcn_s_sleep(1); }
						xchg_order();
						const uint4* r = xq.rec(best, slot);
						const uint4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4];
						xchg_order();
						__hip_atomic_store(sq, p + xq.cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						ray.o = mk<Real>(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
						ray.d = mk<Real>(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
						tv.best_t = __uint_as_float(q1.z); tv.best_prim = (int32_t)q1.w;
						T = mk<Real>(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
						L = mk<Real>(__uint_as_float(q2.w), __uint_as_float(q3.x), __uint_as_float(q3.y));
						rng.state = q3.z; rng.inc = q3.w;
						accum_index = q4.x; depth = (int32_t)q4.y; sample_slot = q4.z;
						state = kShade;
					}
					xchg_order();
					if (lane == 0) atomicAdd(xq.space() + best, want);
				}
			}
		}
		phase_end<kCount>(rg, kPhXchg, t_ph);
		t_ph = phase_begin<kCount>();
		// ---- shading ---------------------------------------------------------------------
		const unsigned long long m_shade = __ballot(state == kShade);
		const bool any_shade = m_shade != 0ull;
		w_rays += (uint32_t)__popcll(m_shade);
		if (state == kShade) {
			region_tick<kCount>(rg, kRegShade);
			if constexpr (kCount && kTally) {  // which arm of the shading code each lane takes: the arms run one after the other
				if (tv.best_prim < 0) region_tick<kCount>(rg, kRegSky);
				else {
					const int m = (prims[tv.best_prim].id >> kMatShift) & 3;
					if (m == RTHIP_DIELECTRIC) region_tick<kCount>(rg, kRegGlass);
					else if (m == RTHIP_METAL) region_tick<kCount>(rg, kRegMetal);
					else region_tick<kCount>(rg, kRegLambert);
				}
			}
			RayT<Real> next;
			if (shade_ray<Real>(ray, tv.best_t, tv.best_prim, prims, mats, A.sky_mode, rng, T, L, depth, &next)) {
				ray.o = next.o; ray.d = next.d;
				state = kFresh;
			} else {
				if (kSamples && A.samples_out) {
					Real* so = A.samples_out + (size_t)sample_slot * 3;
					so[0] = L.x; so[1] = L.y; so[2] = L.z;
				}
				if constexpr (kAccLds) {
					unsigned long long* acc = acc_tile + (accum_index & 63u);
					atomicAdd(acc, to_fixed<Real>(Fp<Real>::sqrt(L.x)));
					atomicAdd(acc + kTilePixels, to_fixed<Real>(Fp<Real>::sqrt(L.y)));
					atomicAdd(acc + 2 * kTilePixels, to_fixed<Real>(Fp<Real>::sqrt(L.z)));
				} else {
					accumulate_sample<Real>(A.accum, accum_index, L);
				}
				state = kIdle;
			}
		}
		if constexpr (!kXchg) break;
		else {
			// another step?  Only after one that shaded something (so that a queue nobody can take from ends
			// the loop), with free lanes and a queue that holds enough for them.
			if (!any_shade) break;
			const int32_t n_free = (int32_t)__popcll(__ballot(state == kIdle));
			if (n_free < A.xchg_pop_min) break;
			const int4 av4 = xq.peek4(xq.avail());
			if (max(max(av4.x, av4.y), av4.z) < A.xchg_pop_min) break;
			phase_end<kCount>(rg, kPhShade, t_ph);
			t_ph = phase_begin<kCount>();
		}
		}

		phase_end<kCount>(rg, kPhShade, t_ph);
		t_ph = phase_begin<kCount>();
		// ---- regeneration: hand idle lanes the next paths of the wave's pool -------------
		for (;;) {
			unsigned long long need = __ballot(state == kIdle);
			if (need == 0ull) break;
			if ((int32_t)__popcll(need) < A.regen_min && need != ~0ull) break;  // wait for more idle lanes: camera rays for few lanes waste the wave
			if (pool_pos == pool_end && (!kBatchRegen || spare_pos == spare_end)) {
				if (queue_dry) break;
				if constexpr (kAccLds) {
					if (need != ~0ull) break;  // paths of the current item are still in flight: the tile is theirs
					flush_tile();
				}
				bool got = false;
				while (queue_hop < kQueues) {
					const uint32_t q = (home_queue + queue_hop) % kQueues;
					const uint32_t begin = (uint32_t)((unsigned long long)A.n_items * q / kQueues);
					const uint32_t end = (uint32_t)((unsigned long long)A.n_items * (q + 1u) / kQueues);
					uint32_t it = 0;
					if (lane == 0) it = atomicAdd(A.queue + q * 16u, 1u);
					it = (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
					if (it < end - begin) { item = begin + it; got = true; break; }
					queue_hop++;
				}
				if (!got) { queue_dry = true; break; }
				pool_pos = 0; pool_end = pool_size;
				if constexpr (kAccLds) {  // the item's tile (work items of LDS-resident scenes run in plain tile order)
					const uint32_t tile = (uint32_t)A.part_index + (item / A.n_chunks) * (uint32_t)A.part_count;
					acc_dst = A.accum + (size_t)tile * (kTilePixels * kAccumChannels);
				}
			}
			if constexpr (kBatchRegen) {
				// Camera rays are made 64 at a time, by all lanes together, and parked one per lane (the
				// "spares"); idle lanes then pick theirs up from whichever lane holds it (ds_bpermute).
				// Making them on demand ran ray generation, a sixth of the kernel's arithmetic, on the
				// 16-30 lanes that happened to be idle.
				// Where the launch asks for it (kPacket), a batch goes further before it is handed out: its rays
				// walk the tree together (packet_trace) and are shaded together, all 64 lanes at work; paths that
				// end with their first ray (sky, absorbed, depth limit) add their sample right there, and the
				// spares that remain are the *second* rays of their paths, moved up so that they stand in a row.
#ifdef RT_PACKET_ONLY
				constexpr bool kFirstOnly = kPacket;
#else
				constexpr bool kFirstOnly = false;
#endif
				const bool first_hit = kPacket && (kFirstOnly || (A.packet != 0 && A.max_depth > 0));
				if (spare_pos == spare_end) {
					const uint32_t n = min(64u, pool_end - pool_pos);
					sp_accum = 0xFFFFFFFFu;  // no path: outside the image, or a sample beyond spp_count
					sp_item = item; sp_base = pool_pos;
					bool sp_valid = false;
					PathId id;
					Rng g;
					g.state = 0; g.inc = 1;
					if (lane < n && decode_path(item, pool_pos + lane, A.n_chunks, A.chunk, A.spp_count, A.part_index, A.part_count, A.tiles_x, A.width, A.height, &id, kLds != 0 ? 0u : A.block_order, first_hit && A.pixel_major != 0)) {
						g.init(A.seed, id.py * (uint32_t)A.width + id.px, A.spp_begin + id.s_local);
						const RayT<Real> r = camera_ray<Real>(A.cam, (Real)id.px, (Real)id.py, g, true);
						sp_o = r.o; sp_d = r.d;
						sp_valid = true;
						if constexpr (!kLeanSpares && !kFirstOnly) {
							sp_state = g.state; sp_inc = g.inc;
							sp_accum = id.accum_index;
							sp_slot = (id.s_local * (uint32_t)A.height + id.py) * (uint32_t)A.width + id.px;
						}
						region_tick<kCount>(rg, kRegStart);
					}
					pool_pos += n;
					spare_pos = 0; spare_end = n;
					if constexpr (kPacket) {
						if (first_hit) {
							const unsigned long long t_pk = phase_begin<kCount>();
							Trav<Real> pt;
							const TravRay<Real> pr = prepare<Real>(RayT<Real>{sp_o, sp_d});
							if constexpr (kLds != 0) {
								typedef const uint32_t* Words;
								packet_trace<Real, kCount, Words>(pt, pr, sp_valid, reinterpret_cast<Words>(nodes), reinterpret_cast<Words>(prims), reinterpret_cast<Words>(gates),
								                                  A.big_leaf, A.start, tally, rg);
							} else {
								// The scene is in global memory and is only read: through the constant address space a
								// load from a wave-uniform address is a scalar load (one request per wave to the scalar
								// cache instead of 64 lane requests to the vector L1, whose request rate is what bounds
								// these scenes).  The packet walks the full nodes: nothing to decode.
								packet_trace<Real, kCount, ConstWords>(pt, pr, sp_valid, (ConstWords)A.scene, (ConstWords)(A.scene + A.prims_off), (ConstWords)(A.scene + A.gates_off),
								                                       A.big_leaf, A.start, tally, rg);
							}
							bool live = false;
							unsigned long long done_r = 0ull, done_g = 0ull, done_b = 0ull;   // the sample of a path that ends here
							uint32_t done_pixel = 0xFFFFFFFFu;                                 // this lane's place in the accumulation buffer
							{
								const uint32_t n_valid = (uint32_t)__popcll(__ballot(sp_valid));
								w_paths += n_valid; w_rays += n_valid;
							}
							if (sp_valid) {
								// (pixel and random stream are worked out again here rather than kept in registers through the walk)
								decode_path(sp_item, sp_base + lane, A.n_chunks, A.chunk, A.spp_count, A.part_index, A.part_count, A.tiles_x, A.width, A.height, &id, kLds != 0 ? 0u : A.block_order, A.pixel_major != 0);
								g.init(A.seed, id.py * (uint32_t)A.width + id.px, A.spp_begin + id.s_local);
								const int draws = A.cam.defocus_angle > Real(0) ? 4 : 2;  // camera_ray: two for the jitter, two for the lens
								for (int k = 0; k < draws; k++) g.skip();
								V3<Real> T1 = mk<Real>(1, 1, 1), L1 = mk<Real>(0, 0, 0);
								int32_t depth1 = A.max_depth;
								RayT<Real> next;
								if (shade_ray<Real>(pr, pt.best_t, pt.best_prim, prims, mats, A.sky_mode, g, T1, L1, depth1, &next)) {
									live = true;
									sp_o = next.o; sp_d = next.d;   // T and L of a path after one hit follow from the material: the lane that takes the ray reads them there
								} else {
									if (kSamples && A.samples_out) {
										Real* so = A.samples_out + ((size_t)(id.s_local * (uint32_t)A.height + id.py) * (uint32_t)A.width + id.px) * 3;
										so[0] = L1.x; so[1] = L1.y; so[2] = L1.z;
									}
									done_r = to_fixed<Real>(Fp<Real>::sqrt(L1.x)); done_g = to_fixed<Real>(Fp<Real>::sqrt(L1.y)); done_b = to_fixed<Real>(Fp<Real>::sqrt(L1.z));
								}
								done_pixel = id.accum_index;
							}
							// Where the batch runs pixel by pixel, the lanes of one pixel stand side by side: their samples are
							// summed along the wave first (integer sums: any order gives the same words) and the first lane
							// of each pixel adds the total, instead of up to 64 lanes hitting one address.
							for (int off = 1; off < 64 && A.pixel_major != 0; off <<= 1) {   // (sample by sample, every lane has a pixel of its own)
								const uint32_t other = (uint32_t)__shfl_down((int)done_pixel, off, 64);
								const unsigned long long r2 = __shfl_down(done_r, off, 64), g2 = __shfl_down(done_g, off, 64), b2 = __shfl_down(done_b, off, 64);
								if (lane + (uint32_t)off < 64u && other == done_pixel) { done_r += r2; done_g += g2; done_b += b2; }
							}
							{
								const uint32_t before = (uint32_t)__shfl_up((int)done_pixel, 1, 64);
								const bool head = A.pixel_major == 0 || lane == 0u || before != done_pixel;
								if (head && done_pixel != 0xFFFFFFFFu && (done_r | done_g | done_b) != 0ull) {
									if constexpr (kAccLds) {
										unsigned long long* acc = acc_tile + (done_pixel & 63u);
										atomicAdd(acc, done_r); atomicAdd(acc + kTilePixels, done_g); atomicAdd(acc + 2 * kTilePixels, done_b);
									} else {
										unsigned long long* acc = A.accum + (size_t)(done_pixel >> 6) * (kTilePixels * kAccumChannels) + (done_pixel & 63u);
										atomicAdd(acc, done_r); atomicAdd(acc + kTilePixels, done_g); atomicAdd(acc + 2 * kTilePixels, done_b);
									}
								}
							}
							// the rays that go on move to the front (the others behind them): ds_permute sends, lane by lane
							const unsigned long long m_live = __ballot(live);
							const uint32_t n_live = (uint32_t)__popcll(m_live);
							const int to = (int)(live ? lane_rank(m_live) : n_live + lane_rank(~m_live)) * 4;
							sp_o.x = __builtin_bit_cast(Real, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, sp_o.x)));
							sp_o.y = __builtin_bit_cast(Real, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, sp_o.y)));
							sp_o.z = __builtin_bit_cast(Real, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, sp_o.z)));
							sp_d.x = __builtin_bit_cast(Real, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, sp_d.x)));
							sp_d.y = __builtin_bit_cast(Real, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, sp_d.y)));
							sp_d.z = __builtin_bit_cast(Real, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, sp_d.z)));
							sp_fh_state = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)g.state);
							sp_fh_hit = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)(((uint32_t)pt.best_prim << 6) | lane));   // the primitive hit, and whose pixel this was
							spare_end = n_live;
							phase_end<kCount>(rg, kPhPacket, t_pk);
						}
					}
				}
				const uint32_t want = (uint32_t)__popcll(need), avail = spare_end - spare_pos;
				const uint32_t take = want < avail ? want : avail;
				const uint32_t rk = lane_rank(need);
				const bool mine = state == kIdle && rk < take;
				const int src = mine ? (int)(spare_pos + rk) : (int)lane;  // every lane takes part in the exchange
				const Real ox = __shfl(sp_o.x, src, 64), oy = __shfl(sp_o.y, src, 64), oz = __shfl(sp_o.z, src, 64);
				const Real dx = __shfl(sp_d.x, src, 64), dy = __shfl(sp_d.y, src, 64), dz = __shfl(sp_d.z, src, 64);
				bool taken_first_hit = false;
				if constexpr (kPacket) {
					if (first_hit) {
						taken_first_hit = true;
						const uint32_t st = (uint32_t)__shfl((int)sp_fh_state, src, 64), hit = (uint32_t)__shfl((int)sp_fh_hit, src, 64);
						if (mine) {
							PathId id;
							decode_path(sp_item, sp_base + (hit & 63u), A.n_chunks, A.chunk, A.spp_count, A.part_index, A.part_count, A.tiles_x, A.width, A.height, &id, kLds != 0 ? 0u : A.block_order, A.pixel_major != 0);
							rng.init(A.seed, id.py * (uint32_t)A.width + id.px, A.spp_begin + id.s_local);   // for the stream's increment
							rng.state = st;
							accum_index = id.accum_index;
							sample_slot = (id.s_local * (uint32_t)A.height + id.py) * (uint32_t)A.width + id.px;
							// shade_ray on a path's first hit leaves T = attenuation and L = emission exactly (1 * x and 0 + x)
							const int32_t prim = (int32_t)(hit >> 6);
							const Mat<Real> mat = mats[prim];
							const int material = (prims[prim].id >> kMatShift) & 3;
							T = material == RTHIP_DIELECTRIC ? mk<Real>(1, 1, 1) : ld3(mat.albedo);
							L = material == RTHIP_LAMBERTIAN ? ld3(mat.albedo) * mat.param : mk<Real>(0, 0, 0);
							depth = A.max_depth - 1;
							ray.o = mk<Real>(ox, oy, oz); ray.d = mk<Real>(dx, dy, dz);
							state = kFresh;
						}
					}
				}
				if constexpr (!kFirstOnly) {
				if (!taken_first_hit) {
				uint32_t g_state = 0, g_inc = 1, g_accum = 0xFFFFFFFFu, g_slot = 0;
				if constexpr (!kLeanSpares) {
					g_state = (uint32_t)__shfl((int)sp_state, src, 64); g_inc = (uint32_t)__shfl((int)sp_inc, src, 64);
					g_accum = (uint32_t)__shfl((int)sp_accum, src, 64); g_slot = (uint32_t)__shfl((int)sp_slot, src, 64);
				} else if (mine) {
					PathId id;
					if (decode_path(sp_item, sp_base + (uint32_t)src, A.n_chunks, A.chunk, A.spp_count, A.part_index, A.part_count, A.tiles_x, A.width, A.height, &id, kLds != 0 ? 0u : A.block_order)) {
						Rng g;
						g.init(A.seed, id.py * (uint32_t)A.width + id.px, A.spp_begin + id.s_local);
						const int draws = A.cam.defocus_angle > Real(0) ? 4 : 2;  // camera_ray: two for the jitter, two for the lens
						for (int k = 0; k < draws; k++) g.skip();
						g_state = g.state; g_inc = g.inc;
						g_accum = id.accum_index;
						g_slot = (id.s_local * (uint32_t)A.height + id.py) * (uint32_t)A.width + id.px;
					}
				}
				w_paths += (uint32_t)__popcll(__ballot(mine && g_accum != 0xFFFFFFFFu));
				if (mine && g_accum != 0xFFFFFFFFu) {
					rng.state = g_state; rng.inc = g_inc;
					accum_index = g_accum;
					sample_slot = g_slot;
					T = mk<Real>(1, 1, 1);
					L = mk<Real>(0, 0, 0);
					depth = A.max_depth;
					if (depth > 0) {
						ray.o = mk<Real>(ox, oy, oz); ray.d = mk<Real>(dx, dy, dz);
						state = kFresh;
					} else if (kSamples && A.samples_out) {  // renderer.rs:10-12: depth <= 0 -> black
						Real* so = A.samples_out + (size_t)sample_slot * 3;
						so[0] = so[1] = so[2] = Real(0);
					}
				}
				}
				}
				spare_pos += take;
			} else {
			uint32_t want = (uint32_t)__popcll(need);
			uint32_t avail = pool_end - pool_pos;
			uint32_t take = want < avail ? want : avail;
			if (state == kIdle) {
				uint32_t rk = lane_rank(need);
				if (rk < take) {
					PathId id;
					if (decode_path(item, pool_pos + rk, A.n_chunks, A.chunk, A.spp_count, A.part_index, A.part_count, A.tiles_x, A.width, A.height, &id, kLds != 0 ? 0u : A.block_order)) {
						rng.init(A.seed, id.py * (uint32_t)A.width + id.px, A.spp_begin + id.s_local);
						RayT<Real> r = camera_ray<Real>(A.cam, (Real)id.px, (Real)id.py, rng, true);
						accum_index = id.accum_index;
						sample_slot = (id.s_local * (uint32_t)A.height + id.py) * (uint32_t)A.width + id.px;
						T = mk<Real>(1, 1, 1);
						L = mk<Real>(0, 0, 0);
						depth = A.max_depth;
						n_paths++;
						region_tick<kCount>(rg, kRegStart);
						if (depth > 0) {
							ray.o = r.o; ray.d = r.d;
							state = kFresh;
						} else if (kSamples && A.samples_out) {  // renderer.rs:10-12: depth <= 0 -> black
							Real* so = A.samples_out + (size_t)sample_slot * 3;
							so[0] = so[1] = so[2] = Real(0);
						}
					}
				}
			}
			pool_pos += take;
			}
		}
		phase_end<kCount>(rg, kPhRegen, t_ph);
		if (__ballot(state != kIdle) == 0ull) {
			if constexpr (kXchg) {  // the wave is out of paths: what still waits in the queues is its work too
				const int4 av4 = xq.peek4(xq.avail());
				if (max(max(av4.x, av4.y), av4.z) > 0) continue;
			}
			break;
		}
		t_ph = phase_begin<kCount>();
		// rays made above by shading (next bounce) and by regeneration (camera rays) are set up together
		if (state == kFresh) {
			trav_begin<Real>(tv, stack.bottom());
			if constexpr (kQ) ray = prepare<Real>(RayT<Real>{ray.o, ray.d}, &A.qframe);
			else ray = prepare<Real>(RayT<Real>{ray.o, ray.d});
			// The oversized primitives (ground slab, sun) first: every ray tests them, here the lanes
			// that start a ray do so together instead of one by one in the leaf loops, and the tree
			// below is walked with their hit distance already known.
			if (A.big_leaf != 0) {
				leaf_tests<Real, kCount>(tv, leaf_ray<kQ, Real>(ray), A.big_leaf, prims, gates, tally, rg);
				tv.cur = A.start;
			}
			if constexpr (kRootOnly) state = kShade;
			else state = A.start == kDone ? kShade : kTraverse;  // (a scene that is one leaf: nothing to walk)
		}
		phase_end<kCount>(rg, kPhFresh, t_ph);

		// ---- traversal: until few lanes are left traversing while others wait -------------
		for (;;) {
			if constexpr (kRootOnly) break;   // (no tree: the loop, the stacks and the box-test operands of a ray are not in this kernel)
			unsigned long long m = __ballot(state == kTraverse);
			if (m == 0ull) break;
			uint32_t n_trav = (uint32_t)__popcll(m);
			if ((int32_t)n_trav <= A.trav_threshold) {
				// lanes that shading or regeneration would serve right now (idle lanes only count once
				// there are enough of them for regeneration to accept, so leaving here always makes progress)
				// (with the LDS tile, an exhausted item is refilled only once all its paths are done: nothing to serve before)
				const bool paths_at_hand = !kAccLds || pool_pos != pool_end || (kBatchRegen && spare_pos != spare_end);
				const bool refill = !queue_dry && paths_at_hand && (int32_t)__popcll(__ballot(state == kIdle)) >= A.regen_min;
				if (refill || __ballot(state == kShade) != 0ull) break;
				if constexpr (kXchg) {  // no paths left to start: idle lanes serve the queues
					if (queue_dry && (int32_t)__popcll(__ballot(state == kIdle)) >= A.xchg_pop_min) {
						const int4 av4 = xq.peek4(xq.avail());
						if (max(max(av4.x, av4.y), av4.z) >= A.xchg_pop_min) break;
					}
				}
			}
			if (state == kTraverse) {
				region_tick<kCount>(rg, kRegTravRound);
				t_ph = phase_begin<kCount>();
				if constexpr (kQ) trav_inner_q<Stack, kCount>(tv, ray, reinterpret_cast<const QNode*>(A.scene + A.qnodes_off), stack, tally, rg, A.inner_exit);
				else trav_inner<Real, Stack, kCount>(tv, ray, nodes, stack, tally, rg, A.inner_exit);
				phase_end<kCount>(rg, kPhInner, t_ph);
				t_ph = phase_begin<kCount>();
				if (tv.cur < 0 && tv.cur != kDone) trav_leaf<Real, Stack, kCount>(tv, leaf_ray<kQ, Real>(ray), prims, gates, stack, tally, rg);
				phase_end<kCount>(rg, kPhLeaf, t_ph);
				if (tv.cur == kDone) state = kShade;
			}
		}

	}

	flush_tile();
	// ---- statistics: one atomic per wave ------------------------------------------------
	unsigned long long v[5 + 2 * kRegCount + kPhCount] = {n_paths + (lane == 0 ? w_paths : 0u), lane == 0 ? w_rays : 0u, tally.box, tally.sphere, tally.cube};
	if constexpr (kCount) {
		rg.clk[kPhTotal] = __builtin_amdgcn_s_memtime() - t_kernel;
		for (int k = 0; k < kRegCount; k++) { v[5 + k] = rg.lane[k]; v[5 + kRegCount + k] = rg.wave[k]; }
		for (int k = 0; k < kPhCount; k++) v[5 + 2 * kRegCount + k] = lane == 0 ? rg.clk[k] : 0ull;  // wave-uniform: counted once
	}
	const int nstat = kCount ? 5 + 2 * kRegCount + kPhCount : 2;
	for (int k = 0; k < nstat; k++) {
		unsigned long long s = v[k];
		for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
		if (lane == 0) atomicAdd(A.stats + k, s);
	}
}

// ---------------------------------------------------------------------------------------
// Parity probe: one primary ray per pixel, hit record written out.
// ---------------------------------------------------------------------------------------
template <typename Real>
__global__ __launch_bounds__(kProbeBlock) void probe_kernel(ProbeArgs<Real> A) {
	extern __shared__ __align__(64) char smem[];
	using Stack = StackLds<kProbeBlock>;
	const Stack stack{reinterpret_cast<int*>(smem) + threadIdx.x};
	const Node<Real>* nodes = reinterpret_cast<const Node<Real>*>(A.scene);
	const Prim<Real>* prims = reinterpret_cast<const Prim<Real>*>(A.scene + A.prims_off);
	const Gate<Real>* gates = reinterpret_cast<const Gate<Real>*>(A.scene + A.gates_off);
	uint32_t i = blockIdx.x * kProbeBlock + threadIdx.x;
	if (i >= (uint32_t)A.width * (uint32_t)A.height) return;
	uint32_t py = i / (uint32_t)A.width, px = i - py * (uint32_t)A.width;
	Rng rng;
	rng.init(A.seed, i, A.sample);
	TravRay<Real> ray = prepare<Real>(camera_ray<Real>(A.cam, (Real)px, (Real)py, rng, A.jitter != 0));
	Trav<Real> tv;
	trav_begin<Real>(tv, stack.bottom());
	Tally tally{0, 0, 0};
	Regions rg{};
	while (tv.cur != kDone) {
		trav_inner<Real, Stack, false>(tv, ray, nodes, stack, tally, rg);
		if (tv.cur != kDone) trav_leaf<Real, Stack, false>(tv, ray, prims, gates, stack, tally, rg);
	}
	if (tv.best_prim < 0) {
		A.id[i] = -1; A.t[i] = 0.0; A.front[i] = 0;
		A.normal[3 * (size_t)i] = A.normal[3 * (size_t)i + 1] = A.normal[3 * (size_t)i + 2] = 0.0;
		return;
	}
	const Prim<Real>& pr = prims[tv.best_prim];
	if constexpr (std::is_same<Real, float>::value) {
		// BVH::trace (src/bvh.rs:133-139) for the winning primitive in the reference's own arithmetic:
		// which primitive is closest is decided in f32, how far it is and which way its surface faces
		// is then one f64 sphere or cube test (vector f64 runs at half the f32 rate: one test per
		// ray is nothing).  If the f64 ray misses that primitive (a silhouette pixel), the f32 record stands.
		if (A.scene64) {
			Rng rng64;
			rng64.init(A.seed, i, A.sample);
			const TravRay<double> ray64 = prepare<double>(camera_ray<double>(A.cam64, (double)px, (double)py, rng64, A.jitter != 0));
			const Prim<double>& p64 = reinterpret_cast<const Prim<double>*>(A.scene64 + A.prims_off64)[tv.best_prim];
			const bool sphere = ((p64.id >> kKindShift) & 1) == RTHIP_SPHERE;
			double t64 = 0.0;
			const bool ok = sphere ? hit_sphere<double>(p64, ray64, &t64) : hit_cube<double>(p64, ray64, &t64);
			if (ok && t64 > 0.001 && t64 < Fp<double>::max_finite()) {
				const Hit<double> h64 = make_hit<double>(p64, sphere, ray64, t64);
				A.id[i] = h64.id; A.t[i] = t64; A.front[i] = h64.front ? 1 : 0;
				A.normal[3 * (size_t)i] = h64.n.x; A.normal[3 * (size_t)i + 1] = h64.n.y; A.normal[3 * (size_t)i + 2] = h64.n.z;
				return;
			}
		}
	}
	Hit<Real> h = make_hit<Real>(pr, ((pr.id >> kKindShift) & 1) == RTHIP_SPHERE, ray, tv.best_t);
	A.id[i] = h.id; A.t[i] = (double)tv.best_t; A.front[i] = h.front ? 1 : 0;
	A.normal[3 * (size_t)i] = (double)h.n.x; A.normal[3 * (size_t)i + 1] = (double)h.n.y; A.normal[3 * (size_t)i + 2] = (double)h.n.z;
}

#undef RT_FN
}  // namespace rthip
