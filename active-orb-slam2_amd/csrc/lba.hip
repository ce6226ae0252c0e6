// placeholder, replaced below
#include "aos2_common.h"
