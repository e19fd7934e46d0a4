"""Drop-in for the vendored `simple_knn` extension (submodules/simple-knn)."""
