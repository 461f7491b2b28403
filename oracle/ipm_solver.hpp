#pragma once
#include "nlp_model.hpp"
namespace orc {
struct IpmOptions { int max_iter = 500; double ref_tol = 1e-3; };
struct IpmResult { int status = -2; int iters = 0; double kkt_error = 0, constr_viol = 0, objective = 0, mu = 0; int n_factor = 0, N = 0, bandwidth = 0; };
inline IpmResult ipm_solve(Problem&, const IpmOptions&) { return IpmResult(); }
}
