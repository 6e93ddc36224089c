"""Synthetic parameter draws for the tests: thin aliases of the package's vectorised host priors."""
from bayesflow_nddms_amd.priors import (alpha_ns_prior_matrix as alpha_ns_prior,  # noqa: F401
                                        basic_prior_matrix as basic_prior, single_prior_matrix as single_prior)
