// device_step.cpp -- which device the engine uses when nobody ever called gtnx_set_device: the one the
// calling thread already has with HIP (a rank that only did torch.cuda.set_device(k)), not device 0 -- and a
// hipSetDevice made behind the engine's back afterwards (torch.cuda.device(...)) must not leave its launches on the
// wrong device: the engine tells HIP its device on every entry instead of trusting a cache.
//   NULLHIP_DEVICES=8 NULLHIP_ZERO=1 LD_PRELOAD=tools/nullhip/_bin/libnullhip.so tools/nullhip/_bin/device_step
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>

#include "gtn/gtn.h"

using namespace gtn;

int main() {
  auto counter = [](const char* name, int d) -> long {
    auto fn = reinterpret_cast<long (*)(int)>(dlsym(RTLD_DEFAULT, name));
    return fn ? fn(d) : -1;
  };
  if (counter("nullhip_launches", 0) < 0 || gtnx_device_count() < 8) {
    std::printf("device_step needs NULLHIP_DEVICES=8 LD_PRELOAD=tools/nullhip/_bin/libnullhip.so\n");
    return 2;
  }
  int failures = 0;
  auto expect = [&](bool ok, const char* what) {
    if (!ok) {
      ++failures;
      std::printf("FAILED: %s\n", what);
    }
  };
  if (hipSetDevice(5) != hipSuccess) return 2;  // what torch.cuda.set_device(5) is
  float* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(float) * 40 * 8) != hipSuccess) return 2;
  auto once = [&] {
    Graph e = linearGraph(40, 8);
    e.setWeights(dev);
    Graph s = forwardScore(e);
    return s.item();
  };
  (void)once();
  int d = -1, hd = -1;
  detail::check(gtnx_get_device(&d));
  (void)hipGetDevice(&hd);
  expect(d == 5, "the engine's device is the one the thread had with HIP (5), not 0");
  expect(hd == 5, "the calling thread is still on its device");
  expect(counter("nullhip_launches", 5) > 0 && counter("nullhip_launches", 0) == 0, "launches went to device 5, none to device 0");
  // behind the engine's back
  if (hipSetDevice(2) != hipSuccess) return 2;
  const long before5 = counter("nullhip_launches", 5), before2 = counter("nullhip_launches", 2);
  (void)once();
  expect(counter("nullhip_launches", 5) > before5 && counter("nullhip_launches", 2) == before2,
         "after a hipSetDevice(2) by somebody else the engine still launches on ITS device");
  for (int k = 0; k < 8; ++k) expect(counter("nullhip_wrong_device_launches", k) == 0, "no launch on a stream of a device that is not current");
  std::printf("%s\n", failures ? "DEVICE_FAIL" : "DEVICE_OK");
  return failures ? 1 : 0;
}
