// Stand-ins for what the library's host-only translation units take from the main one, so that they
// can be built alone (Makefile: asan).  Not part of libldpc_hip.so.
#include <cstdio>

static thread_local char g_text[512] = "";

extern "C" void ldpc_set_error_(const char *text) { std::snprintf(g_text, sizeof(g_text), "%s", text); }
extern "C" const char *ldpc_last_error(void) { return g_text; }
