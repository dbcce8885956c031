raise ImportError("tests/_no_torch: torch is made unimportable for this process on purpose")
