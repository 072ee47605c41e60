/*
 * fastpm_hip_binding.h -- what the translation units that are compiled INSIDE libfastpm (gravity_hip.c, factors_hip.c,
 * store_hip.c, transfer_hip.c: the reference's real structs) share: the GPU plan of a PM, and whether the store columns
 * are device-resident.  Included after the reference's headers.
 */
#ifndef FASTPM_HIP_BINDING_H
#define FASTPM_HIP_BINDING_H

#include <fastpm_hip.h>

/* gravity_hip.c: the plan cache keyed on PM * (made at the first use of that PM) */
fpmhip_plan * fastpm_hip_plan_for(PM * pm);
/* the plan / PM of the latest force call: fastpm_kick_store, fastpm_drift_store and fastpm_store_wrap carry no PM
 * (factors.c:175-197, 373-392; store.c:446-475).  NULL before the first force. */
fpmhip_plan * fastpm_hip_current_plan(void);
PM * fastpm_hip_current_pm(void);
/* the exchanges of that PM's process mesh (a fastpm_hip_transport, fastpm_slab_hip.h; NULL on one rank) */
const void * fastpm_hip_current_transport(void);
/* ... of ANY PM (its plan and transport made at the first use): pm_2lpt_solve runs on the IC mesh, not the force mesh */
const void * fastpm_hip_transport_for(PM * pm);
/* 1 when factors_hip.o is linked in and FASTPM_HIP_RESIDENT is not 0: columns stay on the device between the calls */
int fastpm_hip_resident_enabled(void);

/* store_hip.c: what host code around the replaced functions calls ("resident store") */
void fastpm_hip_store_sync(FastPMStore * p, FastPMColumnTags attributes);       /* before host code READS columns */
void fastpm_hip_store_touched(FastPMStore * p, FastPMColumnTags attributes);    /* after host code WROTE columns */
void fastpm_hip_store_release(FastPMStore * p);                                 /* fastpm_store_destroy */

#endif
