// comm.hpp — RCCL, bound at run time.  The merge of the GPUs' accumulation buffers is the one
// collective of the path (the reference is a single-process renderer: no counterpart).  RCCL is
// looked up with dlopen when the first communicator is made, so that librthip.so loads on hosts
// without it and, under PyTorch, ends up on the librccl the process already holds.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace rthip {

struct RcclApi {
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	std::string error;  // why the library could not be bound, empty when it is

	// The process-wide binding; nullptr (with *why filled) when RCCL cannot be loaded.
	static const RcclApi* get(std::string* why) {
		static RcclApi api;
		static std::once_flag once;
		std::call_once(once, [] {
			void* h = nullptr;
			for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
				h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
				if (h) break;
			}
			if (!h) { api.error = std::string("cannot load librccl: ") + dlerror(); return; }
			auto sym = [&](const char* n) -> void* {
				void* p = dlsym(h, n);
				if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + n;
				return p;
			};
			api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
			api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
			api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
			api.Reduce = reinterpret_cast<decltype(api.Reduce)>(sym("ncclReduce"));
			api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
			api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
		});
		if (!api.error.empty()) { if (why) *why = api.error; return nullptr; }
		return &api;
	}
};

}  // namespace rthip
