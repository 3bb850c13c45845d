"""TEST INFRASTRUCTURE: CPU oracle of the overlap hot path (see flye_oracle.cpp)."""
