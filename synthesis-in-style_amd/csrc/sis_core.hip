// Library-wide state of libsis_hip.so: version and the thread-local error text.
#include "sis_common.h"

thread_local char sis_err_buf[512] = "";

extern "C" int sis_version(void) { return 1000; }
extern "C" const char* sis_last_error(void) { return sis_err_buf; }
