// TEST BUILD ONLY (tests/faultlib/Makefile): defines the fault points of csrc/eg_hip.hip.  The shipped libeg_hip.so is compiled without
// this header; there EG_FAULT_POINT(name) is the constant false and no switch exists.  Here a fault point fires when the environment
// variable EG_TEST_FAIL_<name> is set (e.g. EG_TEST_FAIL_after_fork: engine_verify_device returns an error with the first chunk's
// kernels queued on the work sets' streams).
#pragma once
#include <cstdlib>
#define EG_FAULT_POINT(name) (getenv("EG_TEST_FAIL_" #name) != nullptr)
