// comm.cpp -- collectives between the GPUs ONE process drives (include/gtn_amd.h: gtnx_comm_*), over RCCL / xGMI.
//
// The path shards by utterance and needs no data-path collective (parallelMap has no cross-task
// communication, parallel_map.h:167-179).  What a host that drives several devices gathers afterwards is small:
// the per-utterance losses (all-gather of B / G floats per device) and, for ASG, the gradient of the ONE
// transitions graph every utterance shares (criterion_test.cpp:289-305: the sum over utterances -- per-device
// partial sums, then an all-reduce of C^2 + C floats).  Both run here as RCCL group calls, one communicator per
// device, each enqueued on its device's engine stream (runtime.h: one context per device), so they order with the
// kernels that produce their inputs and nothing waits on the host.
// librccl is looked up at first use (dlsym of what the process already has -- torch loads it -- else
// dlopen("librccl.so")): the engine itself does not link it, and a one-device "communicator" needs none.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "runtime.h"

#define GTNX_API extern "C" __attribute__((visibility("default")))

namespace gtnx {
void set_last_error(const std::string& m);  // capi.cpp
namespace {

// the slice of nccl.h this file uses (RCCL keeps NCCL's ABI: rccl/rccl.h)
using ncclComm_t = void*;
constexpr int kNcclFloat = 7;  // ncclFloat32
constexpr int kNcclSum = 0;    // ncclSum
struct Rccl {
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllGather")) {
      h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return;
    }
    auto sym = [&](const char* n) { return dlsym(h, n); };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.AllGather && r.AllReduce;
  });
  return r;
}

void nccl_check(int rc, const char* what) {
  if (rc == 0) return;
  const Rccl& r = rccl();
  throw Error(GTNX_RUNTIME_ERROR, std::string("[gtnx_comm] ") + what + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

struct Comm {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;  // empty: one device, no library needed
};

template <class F>
gtnx_status_t guarded(F&& f) {
  try {
    f();
    return GTNX_OK;
  } catch (const Error& e) {
    set_last_error(e.what());
    return e.status;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return GTNX_RUNTIME_ERROR;
  }
}

}  // namespace
}  // namespace gtnx

using namespace gtnx;

GTNX_API gtnx_status_t gtnx_comm_create(const int* devices, int n, gtnx_comm_t* out) {
  return guarded([&] {
    if (n < 1 || !devices || !out) throw_invalid("[gtnx_comm_create] at least one device");
    auto c = new Comm();
    c->devices.assign(devices, devices + n);
    for (int k = 0; k < n; ++k) {
      if (devices[k] < 0 || devices[k] >= Runtime::device_count()) {
        delete c;
        throw_invalid("[gtnx_comm_create] no such device");
      }
      for (int j = 0; j < k; ++j)
        if (devices[j] == devices[k]) {
          delete c;
          throw_invalid("[gtnx_comm_create] a device appears twice");
        }
      (void)Runtime::of(devices[k]);  // its context (stream, pools) exists from here on
    }
    static const bool force = std::getenv("GTNX_COMM_FORCE_RCCL") != nullptr;  // (tests: the library even for one device)
    if (n > 1 || force) {
      Rccl& r = rccl();
      if (!r.ok) {
        delete c;
        throw Error(GTNX_RUNTIME_ERROR, "[gtnx_comm_create] librccl.so not found: collectives between devices need RCCL");
      }
      c->comms.resize(size_t(n));
      const int rc = r.CommInitAll(c->comms.data(), n, devices);
      if (rc != 0) {
        delete c;
        nccl_check(rc, "ncclCommInitAll");
      }
    }
    *out = reinterpret_cast<gtnx_comm_t>(c);
  });
}

namespace {
// One grouped collective over the communicator's devices: the group is ALWAYS closed and the calling thread is put
// back on its own device, also when a call in between fails -- the first error is kept and thrown after the
// clean-up (a dangling ncclGroupStart would swallow every later RCCL call of the thread).
struct RcclGroup {
  Rccl& r;
  int back;
  int first_rc = 0;
  const char* first_what = nullptr;
  bool open = false;
  explicit RcclGroup(Rccl& lib) : r(lib), back(Runtime::current_device()) {
    note(r.GroupStart(), "ncclGroupStart");
    open = first_rc == 0;
  }
  void note(int rc, const char* what) {
    if (rc != 0 && first_rc == 0) {
      first_rc = rc;
      first_what = what;
    }
  }
  bool ok() const { return first_rc == 0; }
  void close() {
    if (open) {
      open = false;
      note(r.GroupEnd(), "ncclGroupEnd");
    }
    try {
      Runtime::of(back).activate();
    } catch (...) {
    }
  }
  void finish() {
    close();
    if (first_rc != 0) nccl_check(first_rc, first_what);
  }
  ~RcclGroup() { close(); }  // (unwinding from something that is not an RCCL status: an unknown device, ...)
};
}  // namespace

GTNX_API gtnx_status_t gtnx_comm_destroy(gtnx_comm_t h) {
  return guarded([&] {
    Comm* c = reinterpret_cast<Comm*>(h);
    if (!c) return;
    const int back = Runtime::current_device();
    int first_rc = 0;
    for (size_t k = 0; k < c->comms.size(); ++k) {
      try {
        Runtime::of(c->devices[k]).sync();
      } catch (...) {  // (a device that failed: its communicator is still given back)
      }
      const int rc = rccl().CommDestroy(c->comms[k]);
      if (rc != 0 && first_rc == 0) first_rc = rc;
    }
    delete c;
    if (Runtime::initialized()) Runtime::of(back).activate();
    nccl_check(first_rc, "ncclCommDestroy");
  });
}

GTNX_API gtnx_status_t gtnx_comm_size(gtnx_comm_t h, int* n) {
  return guarded([&] {
    if (!h) throw_invalid("[gtnx_comm_size] null communicator");
    *n = int(reinterpret_cast<Comm*>(h)->devices.size());
  });
}

GTNX_API gtnx_status_t gtnx_comm_all_gather_f32(gtnx_comm_t h, const void* const* send, void* const* recv, int64_t count) {
  return guarded([&] {
    Comm* c = reinterpret_cast<Comm*>(h);
    if (!c || !send || !recv || count < 0) throw_invalid("[gtnx_comm_all_gather_f32] bad arguments");
    const size_t n = c->devices.size();
    if (count == 0) return;
    const int back = Runtime::current_device();
    if (c->comms.empty()) {  // one device: its own block is the whole result
      Runtime& rt = Runtime::of(c->devices[0]);
      rt.activate();
      if (recv[0] != send[0]) rt.d2d(recv[0], send[0], sizeof(float) * size_t(count));
      Runtime::of(back).activate();
      return;
    }
    Rccl& r = rccl();
    RcclGroup g(r);
    for (size_t k = 0; k < n && g.ok(); ++k) {
      Runtime& rt = Runtime::of(c->devices[k]);
      rt.activate();
      g.note(r.AllGather(send[k], recv[k], size_t(count), kNcclFloat, c->comms[k], rt.stream()), "ncclAllGather");
    }
    g.finish();
  });
}

GTNX_API gtnx_status_t gtnx_comm_all_reduce_sum_f32(gtnx_comm_t h, void* const* bufs, int64_t count) {
  return guarded([&] {
    Comm* c = reinterpret_cast<Comm*>(h);
    if (!c || !bufs || count < 0) throw_invalid("[gtnx_comm_all_reduce_sum_f32] bad arguments");
    if (count == 0 || c->comms.empty()) return;  // (one device: the sum of one term)
    Rccl& r = rccl();
    RcclGroup g(r);
    for (size_t k = 0; k < c->devices.size() && g.ok(); ++k) {
      Runtime& rt = Runtime::of(c->devices[k]);
      rt.activate();
      g.note(r.AllReduce(bufs[k], bufs[k], size_t(count), kNcclFloat, kNcclSum, c->comms[k], rt.stream()), "ncclAllReduce");
    }
    g.finish();
  });
}
