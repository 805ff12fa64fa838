// t2o_stamp.hip -- the digest of the sources and flags this library was built from (t2onet_amd/build.py compiles it in;
// the loader refuses a library whose digest differs from the tree's).  Its own translation unit: the only one that is
// recompiled for every change anywhere.
#include "t2onet_hip.h"

#ifndef T2O_SRC_DIGEST
#define T2O_SRC_DIGEST "unstamped"
#endif

extern "C" {
// the tag lets build.py read the digest out of the file without loading it
static const char k_src_digest[] = "t2o-src-digest:" T2O_SRC_DIGEST;
const char* t2o_source_digest(void) { return k_src_digest + 15; }
}
