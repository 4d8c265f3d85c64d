// oracle/_ref build shim: nothing from this header is used by the mapper sources that are compiled.
