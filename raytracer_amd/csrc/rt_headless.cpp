// rt_headless — C++ counterpart of raytracer_headless/src/main.rs:14-73 over the C ABI:
// the host program a maintainer gets by adding `--backend hip` to the reference's headless
// binary.  Same defaults (2560x1440, depth 10, 800 spp, the default scene, the same camera),
// same output (RGB8 PNG, `(v * 255.0) as u8`), same "Rendering took {}s" line; the frame loop
// is one library call instead of the rayon loop.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../include/rthip.h"

namespace {

// ---- minimal PNG writer (RGB8, filter none, zlib deflate) ---------------------------------
uint32_t crc_table[256];
void crc_init() {
	for (uint32_t n = 0; n < 256; n++) {
		uint32_t c = n;
		for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
		crc_table[n] = c;
	}
}
uint32_t crc_update(uint32_t c, const uint8_t* p, size_t n) {
	for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
	return c;
}
void put32(std::vector<uint8_t>& v, uint32_t x) {
	v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& data) {
	put32(out, (uint32_t)data.size());
	size_t start = out.size();
	out.insert(out.end(), tag, tag + 4);
	out.insert(out.end(), data.begin(), data.end());
	put32(out, crc_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu);
}
bool write_png(const std::string& path, const uint8_t* rgb, int w, int h) {
	crc_init();
	std::vector<uint8_t> raw;
	raw.reserve((size_t)h * ((size_t)w * 3 + 1));
	for (int y = 0; y < h; y++) {
		raw.push_back(0);  // filter: none
		raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
	}
	// deflate through zlib, as the image crate does for RgbImage::save (raytracer_headless/src/main.rs:70-72)
	uLongf zlen = compressBound((uLong)raw.size());
	std::vector<uint8_t> z(zlen);
	if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
	z.resize(zlen);
	std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	std::vector<uint8_t> ihdr;
	put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
	ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
	chunk(out, "IHDR", ihdr);
	chunk(out, "IDAT", z);
	chunk(out, "IEND", {});
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) return false;
	bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
	std::fclose(f);
	return ok;
}

int usage() {
	std::fprintf(stderr,
	             "usage: rt_headless --backend hip [--scene default|sample|simple|glass|metal|wallpaper|synthetic:N]\n"
	             "       [--width W] [--height H] [--spp N] [--max-depth D] [--seed S] [--scene-seed S] [--precision f32|f64]\n"
	             "       [--sky reference|wallpaper] [--device I] [--chunk-spp N] [--out output.png]\n"
	             "       [--gpus N] [--split spp|tile] [--devices I,J,...]   several GPUs from this one process: every GPU renders its\n"
	             "       share (its slice of the samples, or its 8x8 tiles), the shares are summed on the first GPU\n");
	return 2;
}

}  // namespace

int main(int argc, char** argv) {
	// raytracer_headless/src/main.rs:15-18
	int width = 2560, height = 1440, max_depth = 10, device = 0;
	uint32_t samples = 800, seed = 0x5EED0001u, scene_seed = 0x5EED0001u, chunk_spp = 100;
	std::string backend = "hip", scene = "default", out = "output.png", precision = "f32", sky = "reference", split = "spp", device_list;
	int gpus = 1;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
		const char* v = nullptr;
		if (a == "--backend" && (v = next())) backend = v;
		else if (a == "--scene" && (v = next())) scene = v;
		else if (a == "--width" && (v = next())) width = std::atoi(v);
		else if (a == "--height" && (v = next())) height = std::atoi(v);
		else if (a == "--spp" && (v = next())) samples = (uint32_t)std::strtoul(v, nullptr, 0);
		else if (a == "--max-depth" && (v = next())) max_depth = std::atoi(v);
		else if (a == "--seed" && (v = next())) seed = (uint32_t)std::strtoul(v, nullptr, 0);
		else if (a == "--scene-seed" && (v = next())) scene_seed = (uint32_t)std::strtoul(v, nullptr, 0);
		else if (a == "--precision" && (v = next())) precision = v;
		else if (a == "--sky" && (v = next())) sky = v;
		else if (a == "--device" && (v = next())) device = std::atoi(v);
		else if (a == "--chunk-spp" && (v = next())) chunk_spp = (uint32_t)std::strtoul(v, nullptr, 0);
		else if (a == "--out" && (v = next())) out = v;
		else if (a == "--gpus" && (v = next())) gpus = std::atoi(v);
		else if (a == "--split" && (v = next())) split = v;
		else if (a == "--devices" && (v = next())) device_list = v;
		else return usage();
	}
	if (backend != "hip") {
		std::fprintf(stderr, "rt_headless: only --backend hip is built here; the CPU backend is the reference's own raytracer_headless\n");
		return 2;
	}
	if (samples == 0 || width < 1 || height < 1 || chunk_spp == 0 || gpus < 1 || gpus > 64 || (split != "spp" && split != "tile")) return usage();
	// the GPU of every share: --devices as given (an ordinal may repeat: shares then take turns on that GPU), else device, device + 1, ...
	std::vector<int> devices;
	for (size_t pos = 0; pos < device_list.size();) {
		size_t comma = device_list.find(',', pos);
		if (comma == std::string::npos) comma = device_list.size();
		devices.push_back(std::atoi(device_list.substr(pos, comma - pos).c_str()));
		pos = comma + 1;
	}
	if (devices.empty()) for (int k = 0; k < gpus; k++) devices.push_back(device + k);
	if ((int)devices.size() != gpus) {
		if (!device_list.empty() && gpus == 1) gpus = (int)devices.size();
		else return usage();
	}

	int64_t n = 0;
	if (rthip_scene_generate(scene.c_str(), scene_seed, nullptr, 0, &n) != RTHIP_OK) {
		std::fprintf(stderr, "rt_headless: %s\n", rthip_last_error(nullptr));
		return 1;
	}
	std::vector<rthip_object> objects((size_t)n);
	rthip_scene_generate(scene.c_str(), scene_seed, objects.data(), n, &n);

	std::vector<rthip_ctx*> ctxs((size_t)gpus, nullptr);
	auto die = [&](const char* what, rthip_ctx* of = nullptr) {
		std::fprintf(stderr, "rt_headless: %s: %s\n", what, rthip_last_error(of));
		for (rthip_ctx* c : ctxs) if (c) rthip_destroy(c);
		return 1;
	};
	// raytracer_headless/src/main.rs:20-30
	rthip_camera_desc cam;
	std::memset(&cam, 0, sizeof(cam));
	if (scene == "wallpaper") {  // raytracer_gpu/src/main.rs:70: origin, yaw -90, fov 90
		rthip_camera_rotation(-90.0, 0.0, cam.direction);
		cam.fov = 90.0;
	} else {
		cam.origin[0] = 13.0; cam.origin[1] = 1.5; cam.origin[2] = 3.0;
		double len = std::sqrt(13.0 * 13.0 + 1.5 * 1.5 + 3.0 * 3.0);
		cam.direction[0] = -(13.0 / len); cam.direction[1] = -(1.5 / len); cam.direction[2] = -(3.0 / len);
		cam.fov = 20.0;
	}
	cam.focus_dist = 10.0; cam.defocus_angle = 0.6; cam.width = width; cam.height = height;
	for (int k = 0; k < gpus; k++) {  // the scene is replicated: every GPU holds its own copy of the tree
		if (rthip_create(devices[(size_t)k], &ctxs[(size_t)k]) != RTHIP_OK) return die("context");
		if (rthip_scene_upload(ctxs[(size_t)k], objects.data(), n) != RTHIP_OK) return die("scene", ctxs[(size_t)k]);
		if (rthip_camera_set(ctxs[(size_t)k], &cam) != RTHIP_OK) return die("camera", ctxs[(size_t)k]);
	}
	rthip_ctx* ctx = ctxs[0];

	auto start = std::chrono::steady_clock::now();  // main.rs:32
	rthip_render_params p;
	std::memset(&p, 0, sizeof(p));
	p.max_depth = max_depth; p.seed = seed; p.part_count = 1;
	p.precision = precision == "f64" ? RTHIP_F64 : RTHIP_F32;
	p.sky_mode = sky == "wallpaper" ? RTHIP_SKY_WALLPAPER : RTHIP_SKY_REFERENCE;
	uint64_t rays = 0;
	double kernel_ms = 0;
	for (uint32_t done = 0; done < samples; done += chunk_spp) {  // main.rs:43: the sample loop, in launches of chunk_spp
		const uint32_t count = std::min(chunk_spp, samples - done);
		// every GPU its share of this launch, queued without waiting; then all of them are waited for
		std::vector<char> launched((size_t)gpus, 0);
		for (int k = 0; k < gpus; k++) {
			if (split == "tile") { p.part_index = k; p.part_count = gpus; p.spp_begin = done; p.spp_count = count; }
			else {
				const uint32_t lo = (uint32_t)((uint64_t)count * (uint64_t)k / (uint64_t)gpus), hi = (uint32_t)((uint64_t)count * (uint64_t)(k + 1) / (uint64_t)gpus);
				p.part_index = 0; p.part_count = 1; p.spp_begin = done + lo; p.spp_count = hi - lo;
			}
			if (p.spp_count == 0) continue;
			if (rthip_render_accumulate_async(ctxs[(size_t)k], &p) != RTHIP_OK) return die("render", ctxs[(size_t)k]);
			launched[(size_t)k] = 1;
		}
		double slowest = 0;
		for (int k = 0; k < gpus; k++) {
			if (!launched[(size_t)k]) continue;
			if (rthip_synchronize(ctxs[(size_t)k]) != RTHIP_OK) return die("render", ctxs[(size_t)k]);
			rthip_stats s;
			rthip_get_stats(ctxs[(size_t)k], &s);
			rays += s.rays; slowest = std::max(slowest, s.kernel_ms);
		}
		kernel_ms += slowest;
	}
	// the shares are sums of integers: added on the first GPU they give the words one GPU would have produced
	if (gpus > 1 && rthip_merge_local(ctxs.data(), gpus, 0) != RTHIP_OK) return die("merge", ctx);
	double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
	std::printf("Rendering took %gs\n", secs);  // main.rs:59-62
	double paths = (double)width * height * samples;
	std::printf("%.1f Mpaths/s, %.1f Mrays/s (kernel time %.3f s), %lld objects\n", paths / secs / 1e6, (double)rays / secs / 1e6,
	            kernel_ms / 1e3, (long long)n);

	std::vector<uint8_t> rgb((size_t)width * height * 3);
	if (rthip_read_rgb8(ctx, samples, rgb.data()) != RTHIP_OK) return die("readback", ctx);  // main.rs:55,64-69
	for (rthip_ctx* c : ctxs) rthip_destroy(c);
	if (!write_png(out, rgb.data(), width, height)) {  // main.rs:70-72
		std::fprintf(stderr, "failed to save to %s\n", out.c_str());
		return 1;
	}
	return 0;
}
