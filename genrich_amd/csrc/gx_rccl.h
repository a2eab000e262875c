// gx_rccl.h -- the library's own collectives: RCCL over xGMI on device buffers, on the library's stream.
//
// Chromosomes shard across GPUs (the reference's per-chromosome loops Genrich.c:2172, 1729,
// 987), so the hot path needs three tiny genome-wide exchanges and nothing else:
//   1. fragLen / ctrlFrag (calcLambda 1817, calcFactor 1980): all-reduce of 2 x int64 fixed-point parts;
//   2. the BH table (hashPval 300-327 runs over all chromosomes): without a control ONE dense all-reduce of bp-at-V
//      (gx_stats.h), otherwise the range-partitioned exchange of gx_bhx.h (all-reduces of disjoint regions, one
//      all-to-all of {p bits, bp} records as grouped send / recv, one back with the q-values);
//   3. the peak list: gathered by the host program (peak_N numbering, 986 / 925).
// librccl is opened at run time, so a single-GPU run needs no RCCL at all.  One communicator per
// context (= per GPU, one process or thread each); the unique id travels by whatever channel the
// host program has (torch.distributed in bench.py, shared memory between the threads of genrich-amd).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace gxrccl {

struct Api {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclAllGather) allGather = nullptr;
  decltype(&ncclSend) send = nullptr;             // (the range-partitioned BH exchange: all-to-all as grouped send / recv)
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
  decltype(&ncclCommCount) commCount = nullptr;   // (optional: introspection only)
};

// (one host thread per GPU may come here side by side: the tables are filled exactly once)
inline const Api* load(std::string* err) {
  static Api api;
  static std::string why;
  static std::once_flag once;
  std::call_once(once, [&]() {
    // The RCCL that belongs to the HIP runtime this process actually runs on: a Python host may have
    // loaded PyTorch's bundled ROCm (its own libamdhip64 + librccl) before or after this library, and an
    // RCCL build only works on the runtime it ships with.  So look next to the loaded libamdhip64 first.
    std::string beside;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
      beside = info.dli_fname;
      const size_t slash = beside.rfind('/');
      beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
    }
    const std::string names[] = {beside + "librccl.so.1", beside + "librccl.so", "librccl.so.1", "librccl.so",
                                 "/opt/rocm/lib/librccl.so.1"};
    for (const std::string& n : names) {
      api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) {
      why = std::string("cannot open librccl: ") + dlerror();
    } else {
      api.getUniqueId = reinterpret_cast<decltype(api.getUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
      api.commInitRank = reinterpret_cast<decltype(api.commInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
      api.commDestroy = reinterpret_cast<decltype(api.commDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
      api.allReduce = reinterpret_cast<decltype(api.allReduce)>(dlsym(api.handle, "ncclAllReduce"));
      api.allGather = reinterpret_cast<decltype(api.allGather)>(dlsym(api.handle, "ncclAllGather"));
      api.getErrorString = reinterpret_cast<decltype(api.getErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
      api.commCount = reinterpret_cast<decltype(api.commCount)>(dlsym(api.handle, "ncclCommCount"));
      api.send = reinterpret_cast<decltype(api.send)>(dlsym(api.handle, "ncclSend"));
      api.recv = reinterpret_cast<decltype(api.recv)>(dlsym(api.handle, "ncclRecv"));
      api.groupStart = reinterpret_cast<decltype(api.groupStart)>(dlsym(api.handle, "ncclGroupStart"));
      api.groupEnd = reinterpret_cast<decltype(api.groupEnd)>(dlsym(api.handle, "ncclGroupEnd"));
      if (!api.getUniqueId || !api.commInitRank || !api.commDestroy || !api.allReduce || !api.allGather ||
          !api.getErrorString || !api.send || !api.recv || !api.groupStart || !api.groupEnd) {
        why = "librccl lacks an entry point";
        api.handle = nullptr;
      }
    }
  });
  if (!api.handle) {
    if (err) *err = why;
    return nullptr;
  }
  return &api;
}

}  // namespace gxrccl
