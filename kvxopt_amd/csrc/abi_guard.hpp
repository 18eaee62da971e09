// No C++ exception may cross the C ABI of include/kvxhip.h: entry points whose bodies allocate on the host
// (std::vector growth in plan building, extraction, spsolve) run through kvx::guarded().
#pragma once
#include <exception>
#include <new>
#include <string>
#include "../../include/kvxhip.h"

namespace kvx {
void set_last_error(const std::string &s);     // api.cpp

template <class Fn>
int guarded(Fn &&fn) noexcept
{
    try {
        return fn();
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return KVX_ENOMEM;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return KVX_EINVAL;
    } catch (...) {
        set_last_error("unknown C++ exception");
        return KVX_EINVAL;
    }
}
}  // namespace kvx
