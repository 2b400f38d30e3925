/* Calls every function include/rthip.h declares, from plain C (gcc, -lrthip, no Python):
 *   c_abi host   the host-only functions (no GPU needed)
 *   c_abi gpu    all of them, on GPU 0
 * Exits 0 when every status is as expected; prints what failed otherwise. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rthip.h"

static int failures = 0;
#define EXPECT(cond)                                                          \
	do {                                                                      \
		if (!(cond)) { printf("FAILED line %d: %s\n", __LINE__, #cond); failures++; } \
	} while (0)
#define OK(call)                                                                                         \
	do {                                                                                                 \
		int32_t rc_ = (call);                                                                            \
		if (rc_ != RTHIP_OK) { printf("FAILED line %d: %s -> %d (%s)\n", __LINE__, #call, rc_, rthip_last_error(ctx)); failures++; } \
	} while (0)

int main(int argc, char** argv) {
	const int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
	rthip_ctx* ctx = NULL;

	/* ---- host-only ---- */
	int64_t n = 0;
	OK(rthip_scene_generate("glass", 1u, NULL, 0, &n));
	EXPECT(n == 17);
	rthip_object* objs = (rthip_object*)calloc((size_t)n, sizeof(rthip_object));
	OK(rthip_scene_generate("glass", 1u, objs, n, &n));
	EXPECT(rthip_scene_generate("no such scene", 1u, NULL, 0, &n) == RTHIP_ERR_INVALID);
	EXPECT(strlen(rthip_last_error(NULL)) > 0);
	n = 17;
	int32_t ids[17], sizes[17], depth = 0;
	int64_t n_leaves = 0;
	OK(rthip_bvh_leaf_order(objs, n, ids, sizes, &n_leaves, &depth));
	EXPECT(n_leaves >= 4 && depth >= 1);
	EXPECT(rthip_bvh_leaf_order(objs, 0, NULL, NULL, NULL, NULL) == RTHIP_ERR_EMPTY);
	int64_t violations = -1, gates = -1, gated = -1;
	int32_t n_nodes = 0;
	uint64_t checksum = 0;
	OK(rthip_bvh_validate(objs, n, &violations, &n_nodes, &depth, &checksum));
	EXPECT(violations == 0 && n_nodes >= 1 && checksum != 0);
	OK(rthip_bvh_gate_count(objs, n, &gates, &gated));
	EXPECT(gates >= 0 && gated >= 0 && gated <= 5);
	double dir[3];
	rthip_camera_rotation(-90.0, -20.0, dir);
	rthip_camera_desc cam = {{1.0, 1.5, 3.0}, {dir[0], dir[1], dir[2]}, 90.0, 10.0, 0.6, 64, 40};
	double fields[19];
	OK(rthip_camera_fields(&cam, fields));
	EXPECT(fields[0] == 1.0 && fields[18] == 0.6);
	if (!gpu) {
		printf("%s\n", failures ? "host: FAILED" : "host: ok");
		return failures != 0;
	}

	/* ---- on the GPU ---- */
	OK(rthip_create(0, &ctx));
	if (!ctx) { printf("no context: %s\n", rthip_last_error(NULL)); return 2; }
	OK(rthip_set_option(ctx, "chunk", 4.0));
	EXPECT(rthip_set_option(ctx, "no-such-knob", 1.0) == RTHIP_ERR_INVALID);
	OK(rthip_set_stream(ctx, NULL));
	OK(rthip_scene_upload(ctx, objs, n));
	OK(rthip_camera_set(ctx, &cam));
	rthip_render_params p;
	memset(&p, 0, sizeof p);
	p.spp_count = 3; p.max_depth = 10; p.seed = 7u; p.part_count = 1;
	OK(rthip_render_accumulate(ctx, &p));
	rthip_stats st;
	OK(rthip_get_stats(ctx, &st));
	{
		int64_t bad = -1; int32_t dev_nodes = 0, dev_depth = 0;
		OK(rthip_scene_validate(ctx, &bad, &dev_nodes, &dev_depth));
		EXPECT(bad == 0 && dev_nodes == st.n_nodes && dev_depth == st.tree_depth);
	}
	EXPECT(st.paths == 64u * 40u * 3u && st.rays >= st.paths);
	/* refit with the objects where they are: the same tree, the same image (compared below) */
	OK(rthip_scene_refit(ctx, objs, n));
	EXPECT(rthip_scene_refit(ctx, objs, n - 1) == RTHIP_ERR_INVALID);
	{
		int64_t bad = -1;
		OK(rthip_scene_validate(ctx, &bad, NULL, NULL));
		EXPECT(bad == 0);
	}
	void* dptr = NULL;
	uint64_t words = 0;
	OK(rthip_accum_buffer(ctx, &dptr, &words));
	EXPECT(dptr != NULL && words == 8u * 5u * 192u);
	uint64_t* a = (uint64_t*)calloc(words, 8);
	uint64_t* b = (uint64_t*)calloc(words, 8);
	OK(rthip_accum_download(ctx, a, words));
	/* the same samples through the two other render entry points */
	OK(rthip_reset_accumulation(ctx));
	OK(rthip_render_accumulate_async(ctx, &p));
	OK(rthip_synchronize(ctx));
	OK(rthip_accum_download(ctx, b, words));
	EXPECT(memcmp(a, b, words * 8) == 0);
	OK(rthip_reset_accumulation(ctx));
	OK(rthip_render_accumulate_noalloc(ctx, &p));
	OK(rthip_synchronize(ctx));
	OK(rthip_accum_download(ctx, b, words));
	EXPECT(memcmp(a, b, words * 8) == 0);
	OK(rthip_accum_upload(ctx, a, words));
	OK(rthip_set_accum_buffer(ctx, dptr, words));
	OK(rthip_set_accum_buffer(ctx, NULL, 0));
	/* read-back in the three forms */
	double* sums = (double*)calloc(64 * 40 * 3, sizeof(double));
	uint8_t* rgb = (uint8_t*)calloc(64 * 40 * 3, 1);
	uint32_t* xrgb = (uint32_t*)calloc(64 * 40, 4);
	OK(rthip_read_accum(ctx, sums));
	OK(rthip_read_rgb8(ctx, 3, rgb));
	OK(rthip_read_xrgb32(ctx, 3, xrgb));
	int agree = 1;
	double total = 0;
	for (int i = 0; i < 64 * 40; i++) {
		agree = agree && xrgb[i] == (((uint32_t)rgb[3 * i] << 16) | ((uint32_t)rgb[3 * i + 1] << 8) | rgb[3 * i + 2]);
		total += sums[3 * i];
	}
	EXPECT(agree && total > 0);
	/* probes */
	int32_t* id = (int32_t*)calloc(64 * 40, 4);
	double* t = (double*)calloc(64 * 40, 8);
	double* nrm = (double*)calloc(64 * 40 * 3, 8);
	uint8_t* front = (uint8_t*)calloc(64 * 40, 1);
	for (int prec = 0; prec < 3; prec++) OK(rthip_trace_primary(ctx, prec, 1, 7u, 0u, id, t, nrm, front));
	double* smp = (double*)calloc(64 * 40 * 3 * 3, 8);
	p.precision = RTHIP_F64;
	OK(rthip_render_samples(ctx, &p, smp));
	p.precision = RTHIP_F32;
	/* merge: a communicator of one rank, and two contexts summed locally */
	uint8_t uid[128];
	OK(rthip_comm_unique_id(uid));
	OK(rthip_comm_init(ctx, uid, 0, 1));
	OK(rthip_merge_reduce(ctx, 0));
	OK(rthip_merge_wait(ctx, NULL));
	OK(rthip_synchronize(ctx));
	OK(rthip_accum_download(ctx, b, words));
	EXPECT(memcmp(a, b, words * 8) == 0);
	rthip_ctx* other = NULL;
	EXPECT(rthip_create(0, &other) == RTHIP_OK && other);
	if (other) {
		EXPECT(rthip_scene_upload(other, objs, n) == RTHIP_OK && rthip_camera_set(other, &cam) == RTHIP_OK);
		p.spp_begin = 3; p.spp_count = 2;
		EXPECT(rthip_render_accumulate(other, &p) == RTHIP_OK);
		rthip_ctx* both[2] = {ctx, other};
		OK(rthip_merge_local(both, 2, 0));
		/* = five samples rendered by one context */
		OK(rthip_accum_download(ctx, a, words));
		OK(rthip_reset_accumulation(ctx));
		p.spp_begin = 0; p.spp_count = 5;
		OK(rthip_render_accumulate(ctx, &p));
		OK(rthip_accum_download(ctx, b, words));
		EXPECT(memcmp(a, b, words * 8) == 0);
		rthip_destroy(other);
	}
	rthip_destroy(ctx);
	rthip_destroy(NULL);
	printf("%s\n", failures ? "gpu: FAILED" : "gpu: ok");
	return failures != 0;
}
