/*
 * fastpm_slab_mpi.h -- the MPI transport of fastpm_hip_slab_force: the three exchanges of the force step as calls
 * on an MPI communicator, i.e. what libfastpm plugs in with pm->Comm2D (pmpfft.c:117-136: Nproc = {NTask, 1}).
 *
 * Compiled only where an MPI is present (`make mpi` in this directory; the image has MPICH 3.3.2 under
 * /opt/conda).  gpu_aware != 0: the device pointers go to MPI as they are (a GPU-aware MPI build over xGMI / RCCL
 * does the copies).  gpu_aware == 0: every exchange is staged through host buffers the transport owns -- works with
 * any MPI, and is what tests/test_gpu_chost.py runs under `mpiexec -n P` with all ranks sharing the one GPU.
 */
#ifndef FASTPM_SLAB_MPI_H
#define FASTPM_SLAB_MPI_H

#include <mpi.h>

#include "fastpm_slab_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rank / nranks are taken from comm.  plan: the plan whose stream the staging copies are ordered on. */
fastpm_hip_transport *fastpm_hip_mpi_transport_create(MPI_Comm comm, fpmhip_plan *plan, int gpu_aware);
void fastpm_hip_mpi_transport_destroy(fastpm_hip_transport *t);

/* The same transport on RCCL (fastpm_slab_rccl.c): grouped ncclSend / ncclRecv over xGMI for the transposes, halo
 * planes and particle rows, ncclAllReduce for the total mass; comm is used for the bootstrap (ncclUniqueId
 * broadcast, row counts).  One rank per GPU; `device` is this rank's GPU. */
fastpm_hip_transport *fastpm_hip_rccl_transport_create(MPI_Comm comm, int device);
void fastpm_hip_rccl_transport_destroy(fastpm_hip_transport *t);
/* ncclCommCount of the communicator the transport built: the ranks RCCL itself counts (bench.py: comm.rccl_ranks) */
int fastpm_hip_rccl_transport_ranks(const fastpm_hip_transport *t);

#ifdef __cplusplus
}
#endif
#endif
