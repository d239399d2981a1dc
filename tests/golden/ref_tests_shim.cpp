// tests/golden/ref_tests_shim.cpp -- TEST INFRASTRUCTURE, linked into tests/golden/_bin/ref_verify and ref_triangle_geometry (tests/golden/ref_tests.mk).
//
// The reference's tutorials and its verify program use Embree's INTERNAL task scheduler (common/tasking/taskschedulerinternal.h; the symbols are dll_export-ed from
// libembree4.so for them) and rely on rtcNewDevice having created its thread pool as a side effect (kernels/common/device.cpp: Device::Device -> TaskScheduler::create):
// TutorialApplication::initRayStats() calls TaskScheduler::threadCount() right after device creation (tutorials/common/tutorial/tutorial.cpp:616-630).  A library that
// implements the documented C API and nothing else -- this repository's -- has no such side effect, so the programs' own copy of the scheduler (compiled from the
// reference's sources into the test binary) would be used uninitialised.  This file does that one call before main().  None of the reference's sources is changed.
#include "common/tasking/taskschedulerinternal.h"
#include <thread>
namespace {
struct CreateScheduler {
  CreateScheduler() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    if (n > 16) n = 16;                       // (every ray of these programs is one blocking rtcIntersect1 call = one kernel launch: more host threads only contend for the GPU queue)
    embree::TaskScheduler::create(n, false, false);
  }
} g_createScheduler;
}
