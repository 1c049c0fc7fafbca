/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see mco_physics.h).
 *
 * P4 `mj_collision` for the box geoms of the PickAndPlace scene (cube, table top, finger pads).
 * Placeholder in this revision: the free-space (Reach) configs run with enable_contact = 0.
 */
#include "mco_physics.h"

void mco_collision(const mco_model* m, mco_data* d) {
  (void)m;
  d->ncon = 0;
}
