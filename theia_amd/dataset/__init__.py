from .features import FeatureIngest, decode_feature, load_feature_stats  # noqa: F401
