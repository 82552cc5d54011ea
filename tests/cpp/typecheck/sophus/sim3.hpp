// see reference_decls.hpp
#include "../reference_decls.hpp"
