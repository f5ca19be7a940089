// Thread-local error channel of the C-ABI (no exceptions cross the boundary).
#pragma once
namespace dgsct {
void set_error(const char* fmt, ...);
const char* last_error();
bool has_error();
void clear_error();
}  // namespace dgsct
