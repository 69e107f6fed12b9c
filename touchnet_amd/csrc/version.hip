// Build identification for libtouchnet_amd.so (see include/touchnet_amd.h).
extern "C" const char* tn_version(void) { return "touchnet_amd 0.1.0 gfx950"; }
