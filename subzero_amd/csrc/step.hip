// step.hip -- placeholder until the B2 surface lands
#include "context.hpp"
void sz_step_destroy(szContext *) {}
