// cusp/distributed.h -- everything of the one-process-per-GPU layer (no reference equivalent):
//   cusp::distributed::communicator       rank / world from the launcher's environment, TCP bootstrap, RCCL through the C-ABI
//   cusp::distributed::vector<T, Space>   this rank's slice of a sharded vector (cusp::blas all-reduces its inner products)
//   cusp::distributed::csr_matrix<...>    row-block sharded CSR operator + the x exchange (all-gather / halo)
//   cusp::distributed::{ell,dia,coo,hyb}_matrix<...>   the same partition + exchange with the rank's block in another format (r4)
//   cusp::multiply(A, x, y), cusp::krylov::cg(A, x, b[, monitor]), cusp::krylov::bicgstab(A, x, b[, monitor])  overloads for them
//   cusp::distributed::poisson5pt(A, m, n)                           every rank generates its own rows
#pragma once
#include "distributed/bicgstab.h"
#include "distributed/cg.h"
#include "distributed/communicator.h"
#include "distributed/csr_matrix.h"
#include "distributed/matrix.h"
#include "distributed/multiply.h"
#include "distributed/vector.h"
