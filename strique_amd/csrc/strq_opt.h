// Switches of the library (not part of the C ABI): strq::opt(key) is what every former getenv("STRQ_...") reads.
// Order: the options of the context the calling thread is working for (strq_set_option(ctx, key, value)), the process-wide
// table (strq_set_option(NULL, key, value)), the environment variable of the same name.  An empty value means "not set".
#pragma once
namespace strq {
const char* opt(const char* key);
}
