"""CPU oracle - test infrastructure only (see oracle/oracle.cpp header).  Never imported by raytracer_amd."""
