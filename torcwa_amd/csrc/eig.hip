// placeholder until the eigensolver lands (next commit)
#include "common.hpp"
extern "C" size_t trx_eig_ws_bytes(int, int, int) { return 0; }
extern "C" int trx_eig(int, void*, void*, void*, int, int, int*, void*, size_t, void*) { return TRX_ERR_UNSUPPORTED; }
