// nxs_guard.hpp -- "never throws across the ABI" (include/nxs_dyn.h, SURVEY 8b: the reference throws std::runtime_error and lets
// terminate() abort the MPI job, FE.cpp:14653; a C caller cannot catch anything).  Every extern "C" entry point of libnxsdyn.so is a
// function-try-block whose handler calls nxs_guard::caught(report): the exception in flight is rethrown here, mapped to a status code
// and handed to the file's own error-text setter -- std::bad_alloc / std::length_error (a std::vector that cannot grow) become
// NXS_ERR_NOMEM, anything else NXS_ERR_INTERNAL with its what().  Nothing in here can throw.
#ifndef NXS_GUARD_HPP
#define NXS_GUARD_HPP

#include <cstdio>
#include <exception>
#include <new>
#include <stdexcept>

#ifndef NXS_ERR_NOMEM
#define NXS_ERR_NOMEM (-6)
#define NXS_ERR_INTERNAL (-7)
#endif

namespace nxs_guard {

// report(code, text) stores the message where the entry point's family keeps it (it may itself fail to allocate: swallowed)
template <class Report>
int caught(const char *entry, Report &&report) noexcept {
    int code = NXS_ERR_INTERNAL;
    char buf[320];
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = NXS_ERR_NOMEM;
        snprintf(buf, sizeof buf, "%s: out of host memory (std::bad_alloc)", entry);
    } catch (const std::length_error &e) {
        code = NXS_ERR_NOMEM;
        snprintf(buf, sizeof buf, "%s: a table would exceed the largest possible size (std::length_error: %s)", entry, e.what());
    } catch (const std::exception &e) {
        snprintf(buf, sizeof buf, "%s: internal error (%s)", entry, e.what());
    } catch (...) {
        snprintf(buf, sizeof buf, "%s: internal error (an exception that is not a std::exception)", entry);
    }
    try { report(code, buf); } catch (...) { }
    return code;
}

}  // namespace nxs_guard

#endif
