// sc_tick_rccl.cpp -- run-time binding of RCCL (see sc_tick_rccl.h).
#include "sc_tick_rccl.h"

#include <dlfcn.h>
#include <cstdlib>
#include <mutex>

namespace sctick {

namespace {
std::once_flag gOnce;
RcclApi gApi{};
bool gOk = false;
std::string gWhy;

void* openRccl(std::string& tried)
{
  // SC_TICK_RCCL_LIB names an explicit library; otherwise the soname (an already loaded copy wins), then ROCm's path
  const char* names[] = { std::getenv("SC_TICK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  for (const char* n : names) {
    if (!n || !*n) continue;
    if (void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)) return h;      // already in the process (e.g. PyTorch's)
  }
  for (const char* n : names) {
    if (!n || !*n) continue;
    if (void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return h;
    const char* err = dlerror();
    tried += std::string(tried.empty() ? "" : "; ") + n + ": " + (err ? err : "not found");
  }
  return nullptr;
}

void bindOnce()
{
  std::string tried;
  void* h = openRccl(tried);
  if (!h) { gWhy = "cannot open RCCL (" + tried + ")"; return; }
  struct Sym { void** slot; const char* name; };
  const Sym syms[] = {
    { (void**)&gApi.GetVersion, "ncclGetVersion" }, { (void**)&gApi.GetUniqueId, "ncclGetUniqueId" },
    { (void**)&gApi.CommInitRank, "ncclCommInitRank" }, { (void**)&gApi.CommDestroy, "ncclCommDestroy" },
    { (void**)&gApi.CommAbort, "ncclCommAbort" }, { (void**)&gApi.CommGetAsyncError, "ncclCommGetAsyncError" },
    { (void**)&gApi.GetErrorString, "ncclGetErrorString" }, { (void**)&gApi.GroupStart, "ncclGroupStart" },
    { (void**)&gApi.GroupEnd, "ncclGroupEnd" }, { (void**)&gApi.Send, "ncclSend" }, { (void**)&gApi.Recv, "ncclRecv" },
    { (void**)&gApi.AllGather, "ncclAllGather" },
  };
  for (const Sym& s : syms) {
    *s.slot = dlsym(h, s.name);
    if (!*s.slot) { gWhy = std::string("RCCL lacks ") + s.name; return; }
  }
  gOk = true;
}
} // namespace

const RcclApi* rccl(std::string* why)
{
  std::call_once(gOnce, bindOnce);
  if (!gOk) { if (why) *why = gWhy; return nullptr; }
  return &gApi;
}

} // namespace sctick
