// oracle/_ref build shim (TEST INFRASTRUCTURE): named by the node's header, nothing of it is used
