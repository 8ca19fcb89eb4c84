// mrt_guard.hpp -- no C++ exception leaves the library (include/mrt.h promises a C boundary).
//
// The host layer allocates through the C++ runtime (staging vectors sized by the caller's counts,
// error strings, worker threads), so std::bad_alloc or std::system_error can be thrown in the
// middle of an entry point.  Every entry point that can meet one is a function-try-block closed
// by MRT_GUARD(setter): the exception becomes a status code (MRT_ERR_NO_MEMORY for a failed
// allocation, MRT_ERR_INVALID otherwise) and, through `setter`, the handle's error message.
#pragma once

#include "mrt.h"

#include <exception>
#include <new>

namespace mrt {

// Call inside a catch (...) handler.  `set(code, message)` stores the message where the entry
// point's mrt_*_last_error finds it; whatever it throws in turn is dropped.
template <typename SetError>
int on_exception(SetError set) noexcept {
    int code = MRT_ERR_INVALID;
    const char* what = "C++ exception";
    try {
        throw;
    } catch (const std::bad_alloc&) {
        code = MRT_ERR_NO_MEMORY;
        what = "out of memory";          // short enough for a std::string to hold without allocating
    } catch (const std::exception& e) {
        what = e.what();                 // (the exception lives as long as the enclosing handler)
    } catch (...) {
    }
    try {
        set(code, what);
    } catch (...) {
    }
    return code;
}

}  // namespace mrt

#define MRT_GUARD(setter)                                                                      \
    catch (...) {                                                                              \
        return mrt::on_exception([&](int code_, const char* what_) { (void)code_; (void)what_; setter; }); \
    }
