// cusp/monitor.h -- iteration monitor with the reference's semantics (cusp/monitor.h:118-250,
// cusp/detail/monitor.inl): stop when ||r|| <= absolute + relative*||b|| or at the iteration limit;
// keeps the residual history.
#pragma once
#include <iomanip>
#include <iostream>
#include <limits>
#include <vector>

#include "blas/blas.h"

namespace cusp {

template <typename ValueType> class monitor {
public:
    typedef ValueType Real;

    template <typename VectorType>
    monitor(const VectorType &b, size_t iteration_limit = 500, Real relative_tolerance = 1e-5, Real absolute_tolerance = 0, bool verbose = false)
        : b_norm(cusp::blas::nrm2(b)), r_norm(std::numeric_limits<Real>::max()), iteration_limit_(iteration_limit), iteration_count_(0),
          relative_tolerance_(relative_tolerance), absolute_tolerance_(absolute_tolerance), verbose(verbose)
    {
        if (verbose) {
            std::cout << "Solver will continue until residual norm " << relative_tolerance << " or reaching " << iteration_limit << " iterations " << std::endl;
            std::cout << "  Iteration Number  | Residual Norm" << std::endl;
        }
        residuals.reserve(iteration_limit);
    }

    void operator++() { ++iteration_count_; }
    bool converged() const { return residual_norm() <= tolerance(); }
    Real residual_norm() const { return r_norm; }
    size_t iteration_count() const { return iteration_count_; }
    size_t iteration_limit() const { return iteration_limit_; }
    Real relative_tolerance() const { return relative_tolerance_; }
    Real absolute_tolerance() const { return absolute_tolerance_; }
    Real tolerance() const { return absolute_tolerance() + relative_tolerance() * b_norm; }
    void set_verbose(bool v = true) { verbose = v; }
    bool is_verbose() { return verbose; }

    template <typename Vector> void reset(const Vector &b)
    {
        b_norm = cusp::blas::nrm2(b);
        r_norm = std::numeric_limits<Real>::max();
        iteration_count_ = 0;
        residuals.resize(0);
    }

    // reference monitor.inl:181-207
    template <typename Vector> bool finished(const Vector &r) { return finished_norm(cusp::blas::nrm2(r)); }

    // same bookkeeping for a solver that already holds ||r|| (the fused device CG gets it for free)
    bool finished_norm(Real norm)
    {
        r_norm = norm;
        residuals.push_back(r_norm);
        if (verbose) std::cout << "       " << std::setw(10) << iteration_count() << "       " << std::setw(10) << std::scientific << residual_norm() << std::endl;
        if (converged()) {
            if (verbose) std::cout << "Successfully converged after " << iteration_count() << " iterations." << std::endl;
            return true;
        }
        if (iteration_count() >= iteration_limit()) {
            if (verbose) std::cout << "Failed to converge after " << iteration_count() << " iterations." << std::endl;
            return true;
        }
        return false;
    }

    Real immediate_rate()
    {
        const size_t n = residuals.size();
        return n < 2 ? Real(0) : residuals[n - 1] / residuals[n - 2];
    }
    Real geometric_rate()
    {
        const size_t n = residuals.size();
        return n < 2 ? Real(0) : std::pow(residuals[n - 1] / residuals[0], Real(1) / Real(n - 1));
    }
    Real average_rate()
    {
        const size_t n = residuals.size();
        if (n < 2) return Real(0);
        Real s = 0;
        for (size_t i = 1; i < n; i++) s += residuals[i] / residuals[i - 1];
        return s / Real(n - 1);
    }

    void print()
    {
        if (iteration_count() == 0) {
            std::cout << "Monitor configured with " << tolerance() << " tolerance and iteration limit " << iteration_limit() << std::endl;
            return;
        }
        if (converged()) std::cout << "Solver converged to " << tolerance() << " tolerance";
        else if (iteration_count() >= iteration_limit()) std::cout << "Solver reached iteration limit " << iteration_limit() << " before converging";
        else throw cusp::runtime_exception("Monitor is in inconsistent state.");
        std::cout << " to (" << residual_norm() << " final residual)" << std::endl;
        std::cout << "Ran " << iteration_count() << " iterations with a final residual of " << r_norm << std::endl;
    }

    std::vector<Real> residuals;

protected:
    Real b_norm, r_norm;
    size_t iteration_limit_, iteration_count_;
    Real relative_tolerance_, absolute_tolerance_;
    bool verbose;
};

} // namespace cusp
