// RCCL bound at run time (dlopen), never at link time: a Python process that has
// imported torch already carries torch's own librccl.so / libamdhip64.so, and a second
// copy of either in the same process is a recipe for mismatched streams.  The caller
// names the library to use (prcg_comm_init's rccl_path).
#pragma once
#include <rccl/rccl.h>

#include <string>

namespace prcg {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;

    // process-wide singleton per path; nullptr + err on failure
    static Rccl* get(const char* path, std::string& err);
};

}  // namespace prcg
