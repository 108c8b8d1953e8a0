"""In-database inference with a model sliced by the query's predicate (role of
the reference's Trails/internal/ml/model_slicing)."""
