// End-of-run gather of the shards' results to one device over RCCL (xGMI between the GPUs of a node).
//
// The reference splits the parameter rows over worker ranks with 2 x MPI_Send / MPI_Recv (src/main.cpp:258-309,
// 357-368) and never gathers: every rank writes its own files (main.cpp:796-797).  Here ONE process drives the GPUs
// of a node; the hillslopes never talk during the integration, and the only exchange of the whole path is this
// gather: 52 bytes per hillslope (N_EQ final states, stiff flag, two step counters) from every shard's device into
// one set of arrays on the root device, as a single group of ncclSend / ncclRecv pairs -- point-to-point, so the
// 7 xGMI links into the root carry 7 peers' shards at once (a ring collective would be per-link bound and there is
// nothing to reduce).
//
// RCCL is loaded on first use (dlopen by soname: a process that already maps an RCCL -- PyTorch's -- shares it,
// like the HIP runtime), so a run that never gathers pays nothing for it and the library has no link-time
// dependency on it.  No fallback: a missing RCCL is an error of the call that asked for it.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

namespace hlm {

struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    void* handle = nullptr;
    std::string error;
};

// the process's RCCL entry points (NULL + *why when the library cannot be loaded)
inline const RcclApi* rccl_api(std::string* why)
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
            api.error = dlerror();
        }
        if (!api.handle) return;
        auto sym = [&](const char* n) {
            void* p = dlsym(api.handle, n);
            if (!p) api.error = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        if (!api.CommInitAll || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.GetErrorString) {
            dlclose(api.handle);
            api.handle = nullptr;
        }
    });
    if (!api.handle) {
        if (why) *why = api.error.empty() ? "librccl.so.1 not found" : api.error;
        return nullptr;
    }
    return &api;
}

// One piece of the gather: `bytes` from `src` on the device of rank `from` into `dst` on the root's device.
struct GatherPiece {
    int from;          // rank (index into the device list) that holds src
    const void* src;
    void* dst;         // on the root's device
    size_t bytes;
};

// One communicator over `devices` (distinct HIP devices; rank i = devices[i]), kept for the lifetime of the object.
class RcclGather {
public:
    RcclGather() = default;
    RcclGather(const RcclGather&) = delete;
    RcclGather& operator=(const RcclGather&) = delete;
    ~RcclGather() { destroy(); }

    // "" on success, else what failed
    std::string init(const std::vector<int>& devices)
    {
        std::string why;
        api_ = rccl_api(&why);
        if (!api_) return "RCCL is not available: " + why;
        devices_ = devices;
        comms_.assign(devices.size(), nullptr);
        const ncclResult_t r = api_->CommInitAll(comms_.data(), (int)devices.size(), devices.data());
        if (r != ncclSuccess) {
            comms_.clear();
            return std::string("ncclCommInitAll failed: ") + api_->GetErrorString(r);
        }
        return "";
    }

    // Enqueue all pieces as ONE group: rank `from` sends on streams[from], the root (rank `root`) receives on
    // streams[root]; a piece the root holds itself goes through the communicator like any other (send to self).
    // Stream-ordered after whatever the streams already hold (the shards' kernels).  "" on success.
    std::string gather(const std::vector<GatherPiece>& pieces, int root, const std::vector<hipStream_t>& streams)
    {
        if (comms_.empty()) return "communicator not initialised";
        ncclResult_t r = api_->GroupStart();
        if (r != ncclSuccess) return std::string("ncclGroupStart failed: ") + api_->GetErrorString(r);
        std::string err;
        for (const GatherPiece& p : pieces) {
            if (p.bytes == 0) continue;
            if (hipSetDevice(devices_[(size_t)p.from]) != hipSuccess) { err = "hipSetDevice failed"; break; }
            r = api_->Send(p.src, p.bytes, ncclUint8, root, comms_[(size_t)p.from], streams[(size_t)p.from]);
            if (r != ncclSuccess) { err = std::string("ncclSend failed: ") + api_->GetErrorString(r); break; }
            if (hipSetDevice(devices_[(size_t)root]) != hipSuccess) { err = "hipSetDevice failed"; break; }
            r = api_->Recv(p.dst, p.bytes, ncclUint8, p.from, comms_[(size_t)root], streams[(size_t)root]);
            if (r != ncclSuccess) { err = std::string("ncclRecv failed: ") + api_->GetErrorString(r); break; }
        }
        r = api_->GroupEnd();   // (always closed, also after an error inside the group)
        if (err.empty() && r != ncclSuccess) err = std::string("ncclGroupEnd failed: ") + api_->GetErrorString(r);
        return err;
    }

    int version() const
    {
        int v = 0;
        if (api_ && api_->GetVersion) (void)api_->GetVersion(&v);
        return v;
    }

    // ranks of the communicator as RCCL counts them (ncclCommCount on the root's handle); 0 if it cannot say
    int nranks() const
    {
        int n = 0;
        if (api_ && api_->CommCount && !comms_.empty() && api_->CommCount(comms_[0], &n) != ncclSuccess) n = 0;
        return n;
    }

    void destroy()
    {
        for (size_t i = 0; i < comms_.size(); ++i)
            if (comms_[i]) {
                (void)hipSetDevice(devices_[i]);
                (void)api_->CommDestroy(comms_[i]);
            }
        comms_.clear();
    }

    const std::vector<int>& devices() const { return devices_; }

private:
    const RcclApi* api_ = nullptr;
    std::vector<int> devices_;
    std::vector<ncclComm_t> comms_;
};

// The process keeps one communicator per device list: ncclCommInitAll takes ~1 s the first time RCCL comes up in a
// process and tens of milliseconds afterwards, a run's gather half a millisecond.  Never destroyed (at exit the HIP
// runtime may already be gone).  *err is set when the communicator cannot be created.
inline RcclGather* rccl_gather_for(const std::vector<int>& devices, std::string* err)
{
    static std::mutex mu;
    static std::vector<RcclGather*>* cache = new std::vector<RcclGather*>();
    std::lock_guard<std::mutex> lk(mu);
    for (RcclGather* g : *cache)
        if (g->devices() == devices) return g;
    RcclGather* g = new RcclGather();
    const std::string e = g->init(devices);
    if (!e.empty()) {
        if (err) *err = e;
        delete g;
        return nullptr;
    }
    cache->push_back(g);
    return g;
}

}  // namespace hlm
