// prost/prost.hpp -- umbrella header of the host C++ API.
#ifndef PROST_PROST_HPP_
#define PROST_PROST_HPP_
#include "prost/backend/backend_admm.hpp"
#include "prost/backend/backend_pdhg.hpp"
#include "prost/linop/blocks.hpp"
#include "prost/linop/linearoperator.hpp"
#include "prost/problem.hpp"
#include "prost/prox/proxes.hpp"
#include "prost/solver.hpp"
#endif
