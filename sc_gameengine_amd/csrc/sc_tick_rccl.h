// sc_tick_rccl.h -- the handful of RCCL entry points the tiled tick needs, resolved at run time.
//
// RCCL is opened with dlopen when the first communicator call arrives, not linked: a single-GPU host never touches
// it, and inside a process that already carries an RCCL (PyTorch bundles its own copy under the same soname) the
// loader hands back that copy instead of bringing a second one into the address space.
#pragma once
#include <rccl/rccl.h>
#include <string>

namespace sctick {

struct RcclApi {
  ncclResult_t (*GetVersion)(int*);
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommAbort)(ncclComm_t);
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*);
  const char*  (*GetErrorString)(ncclResult_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);      // the visible counts ("one tiny all-gather")
};

// nullptr when librccl cannot be opened or lacks a symbol; *why receives the reason
const RcclApi* rccl(std::string* why);

} // namespace sctick
