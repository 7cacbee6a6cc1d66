// tests/stub_cv: see opencv2/core/core.hpp (nothing of this module is named by the reference's headers on the hot path)
#include "../core/core.hpp"
