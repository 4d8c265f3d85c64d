"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/esvo_oracle.h).  Never imported by esvo_amd."""
