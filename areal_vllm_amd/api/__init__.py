"""Public API: configs, allocation modes, IO structs, engine / workflow / reward / environment interfaces."""
