// orbfe_host.h -- internal host-side accessors shared by the translation units of liborbfe.so.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include "../../include/orbfe.h"

// growable device scratch buffer shared by the host-side entry points (matchers, BoW, pose)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (hipMalloc(&p, need) != hipSuccess) return -1;
        bytes = need;
        return 0;
    }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

struct orbfe_match_state;
orbfe_match_state *orbfe_match_state_create();
void orbfe_match_state_destroy(orbfe_match_state *s);

int orbfe_fail(orbfe_context *ctx, int code, const char *fmt, ...);
// Every entry point that takes a (non-const) context holds the context's mutex for its whole duration: the reference's
// Tracking, LocalMapping and LoopClosing threads each construct ORBmatcher objects (src/LocalMapping.cc:215,482,
// src/LoopClosing.cc:275,623) and the shim gives them all the left extractor's context, whose matcher / BoW / database state
// (pinned staging, growable device buffers, grid cache, the stream) is single-user.  Recursive: entry points call each other.
std::recursive_mutex &orbfe_ctx_mutex(orbfe_context *ctx);
#define ORBFE_ENTRY(ctx)                                  \
    std::unique_lock<std::recursive_mutex> orbfe_entry_lock_; \
    if (ctx) orbfe_entry_lock_ = std::unique_lock<std::recursive_mutex>(orbfe_ctx_mutex(ctx))
orbfe_match_state *orbfe_ctx_match_state(orbfe_context *ctx);
hipStream_t orbfe_ctx_stream(orbfe_context *ctx);
int orbfe_ctx_device(const orbfe_context *ctx);
const orbfe_params *orbfe_ctx_params(const orbfe_context *ctx);
const float *orbfe_ctx_scale_factors(const orbfe_context *ctx);
// device-resident frames of the latest extraction call (orbfe_match.hip, orbfe_bow.hip)
struct DeviceConfig;
struct DeviceBuffers;
const DeviceConfig *orbfe_ctx_config(const orbfe_context *ctx);
const DeviceBuffers *orbfe_ctx_buffers(const orbfe_context *ctx);
int orbfe_ctx_slot_count(orbfe_context *ctx, int slot, int *cnt); // keypoints in image slot `slot` of the latest extraction call
unsigned orbfe_ctx_epoch(const orbfe_context *ctx);          // counts the extraction calls enqueued on this context
int orbfe_ctx_wait_foreign_stream(orbfe_context *ctx);      // makes the context's stream wait for the latest extraction call (event, no host wait)

struct orbfe_bow_state;
orbfe_bow_state *orbfe_bow_state_create();
void orbfe_bow_state_destroy(orbfe_bow_state *s);
orbfe_bow_state *orbfe_ctx_bow_state(orbfe_context *ctx);

struct orbfe_pose_state;
orbfe_pose_state *orbfe_pose_state_create();
void orbfe_pose_state_destroy(orbfe_pose_state *s);
orbfe_pose_state *orbfe_ctx_pose_state(orbfe_context *ctx);
const float *orbfe_ctx_inv_sigma2(const orbfe_context *ctx);

// No C++ exception may cross the C ABI (a ctypes / cgo / C caller would abort): every extern "C" function that returns a status is a
// function-try-block closed by this handler.  The per-context lock of ORBFE_ENTRY is a local of the try block, so it is released first.
#include <exception>
#include <new>
#define ORBFE_CATCH(ctxexpr)                                                                                              \
    catch (const std::bad_alloc &) { return orbfe_fail(ctxexpr, ORBFE_ERR_HIP, "out of host memory"); }                   \
    catch (const std::exception &e_) { return orbfe_fail(ctxexpr, ORBFE_ERR_HIP, "host exception: %s", e_.what()); }      \
    catch (...) { return orbfe_fail(ctxexpr, ORBFE_ERR_HIP, "unknown host exception"); }
