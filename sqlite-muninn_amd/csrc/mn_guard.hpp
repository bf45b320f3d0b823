// mn_guard.hpp — the exception barrier of the C-ABI.
//
// The reference's algorithm layer reports memory exhaustion through its return values: hnsw_create → NULL, hnsw_insert /
// hnsw_delete → -1, hnsw_search → 0 (src/hnsw_algo.h:55-79), and hnsw_vtab.c turns those into SQLITE_ERROR / SQLITE_NOMEM
// (src/hnsw_vtab.c:749-752).  The host side of this library is C++ (std::vector / std::string): a std::bad_alloc or
// std::length_error crossing an extern "C" frame would std::terminate the SQLite host.  Every exported function whose body
// can allocate is therefore a function-try-block closed by MN_GUARD_END: the exception becomes the function's error value,
// the thread's last-error string names it, and — where the call may have left a handle half-edited — the handle is marked
// unusable so that later calls fail cleanly instead of serving a torn graph.
//
// Test hook: mn_debug_fault_alloc(n) makes the n-th host allocation made by this library from now on throw std::bad_alloc
// (mn_index.hip replaces operator new inside the library only: hidden visibility).
#pragma once
#include <exception>
#include <new>

#define MN_GUARD_END(SETERR, ONFAIL, ...)                                    \
    catch (const std::bad_alloc &) {                                         \
        ONFAIL;                                                              \
        try {                                                                \
            SETERR("%s: out of host memory", __func__);                      \
        } catch (...) {                                                      \
        }                                                                    \
        return __VA_ARGS__;                                                  \
    }                                                                        \
    catch (const std::exception &e__) {                                      \
        ONFAIL;                                                              \
        try {                                                                \
            SETERR("%s: %s", __func__, e__.what());                          \
        } catch (...) {                                                      \
        }                                                                    \
        return __VA_ARGS__;                                                  \
    }                                                                        \
    catch (...) {                                                            \
        ONFAIL;                                                              \
        try {                                                                \
            SETERR("%s: unknown C++ exception", __func__);                   \
        } catch (...) {                                                      \
        }                                                                    \
        return __VA_ARGS__;                                                  \
    }
#define MN_NOTHING ((void)0)
