#include "prcg_rccl.h"

#include <dlfcn.h>

#include <map>
#include <mutex>

namespace prcg {

Rccl* Rccl::get(const char* path, std::string& err) {
    static std::mutex mu;
    static std::map<std::string, Rccl*> loaded;
    const std::string key = (path && *path) ? path : "librccl.so.1";
    std::lock_guard<std::mutex> lock(mu);
    auto it = loaded.find(key);
    if (it != loaded.end()) return it->second;

    void* lib = dlopen(key.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        err = std::string("dlopen(") + key + ") failed: " + dlerror();
        return nullptr;
    }
    Rccl* r = new Rccl();
    r->lib = lib;
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(lib, name);
        if (!p) { err = std::string("RCCL symbol missing: ") + name; ok = false; }
        return p;
    };
    r->GetUniqueId = reinterpret_cast<decltype(r->GetUniqueId)>(sym("ncclGetUniqueId"));
    r->CommInitRank = reinterpret_cast<decltype(r->CommInitRank)>(sym("ncclCommInitRank"));
    r->CommDestroy = reinterpret_cast<decltype(r->CommDestroy)>(sym("ncclCommDestroy"));
    r->AllReduce = reinterpret_cast<decltype(r->AllReduce)>(sym("ncclAllReduce"));
    r->AllGather = reinterpret_cast<decltype(r->AllGather)>(sym("ncclAllGather"));
    r->Send = reinterpret_cast<decltype(r->Send)>(sym("ncclSend"));
    r->Recv = reinterpret_cast<decltype(r->Recv)>(sym("ncclRecv"));
    r->GroupStart = reinterpret_cast<decltype(r->GroupStart)>(sym("ncclGroupStart"));
    r->GroupEnd = reinterpret_cast<decltype(r->GroupEnd)>(sym("ncclGroupEnd"));
    r->GetErrorString = reinterpret_cast<decltype(r->GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { delete r; return nullptr; }
    loaded[key] = r;
    return r;
}

}  // namespace prcg
