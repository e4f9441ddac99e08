// abi_guard.hpp -- nothing unwinds through the C boundary.
//
// The callers are C: a PostgreSQL backend (lantern_hnsw, which warns about exactly this class of failure at
// lantern_hnsw/src/hnsw/utils.h:22-25 and turns every usearch error string into elog(ERROR): hnsw.c:341-343, scan.c:100,
// build.c:545-551), the Rust server through a C++ bridge that expects error values.  A std::bad_alloc or std::length_error
// from a std::vector growing inside the library would otherwise reach the end of an `extern "C"` frame: std::terminate, a dead
// backend, a crashed postmaster child.  Every `extern "C"` function with a body of its own is therefore a function-try-block
// closed by one of the two macros below: the exception becomes the entry point's error string (static storage) and the function
// returns its "nothing" value (0 / NULL / a zeroed struct).  Locks held by lock_guard are released by the unwinding.
#pragma once
#include <exception>
#include <new>
#include <stdexcept>

namespace lgpu {
inline void abi_fail(const char **e, const char *msg)
{
    if(e) *e = msg;
}
inline void abi_fail(std::nullptr_t, const char *) {}
constexpr const char *kAbiOom = "lantern_gpu: out of host memory";
constexpr const char *kAbiTooLarge = "lantern_gpu: requested size exceeds what can be allocated";
constexpr const char *kAbiException = "lantern_gpu: internal error (C++ exception stopped at the C boundary)";
}  // namespace lgpu

#define LANTERN_ABI_CATCH_(e, RET)                                               \
    catch(const std::bad_alloc &) { lgpu::abi_fail(e, lgpu::kAbiOom); RET; }            \
    catch(const std::length_error &) { lgpu::abi_fail(e, lgpu::kAbiTooLarge); RET; }    \
    catch(const std::exception &) { lgpu::abi_fail(e, lgpu::kAbiException); RET; }      \
    catch(...) { lgpu::abi_fail(e, lgpu::kAbiException); RET; }
#define LANTERN_ABI_CATCH_VOID(e) LANTERN_ABI_CATCH_(e, return)
#define LANTERN_ABI_CATCH(e) LANTERN_ABI_CATCH_(e, return {})
