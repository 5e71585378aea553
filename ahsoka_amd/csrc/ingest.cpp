// placeholder until the ingest lands (next commit): keeps the ABI complete
#include "host_batch.h"
extern "C" int ahsoka_ingest(const char*, const char*, ahsoka_chain_batch**) { return AHSOKA_E_UNSUPPORTED; }
