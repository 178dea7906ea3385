"""TEST INFRASTRUCTURE ONLY -- see oracle/btl_oracle.h.  Not imported by the product package."""
