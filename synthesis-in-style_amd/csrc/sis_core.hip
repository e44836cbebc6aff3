// Library-wide state of libsis_hip.so: version, the thread-local error text and the name of the last convolution kernel.
#include "sis_common.h"

thread_local char sis_err_buf[512] = "";
thread_local const char* sis_kernel_name = "";

extern "C" int sis_version(void) { return 1000; }
extern "C" const char* sis_last_error(void) { return sis_err_buf; }
extern "C" const char* sis_last_kernel(void) { return sis_kernel_name; }
