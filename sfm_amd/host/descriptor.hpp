// Forwarder: the types of the reference header of this name live in sfm_data.hpp here.
#include "sfm_data.hpp"
