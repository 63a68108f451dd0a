// Reference header name kept for source compatibility; the declarations live in matcher_interfaces.hpp.
#include "matcher_interfaces.hpp"
