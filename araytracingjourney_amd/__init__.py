"""MI355X-native software ray-tracing core standing in for the Vulkan RT path of ARayTracingJourney."""
__version__ = "0.1.0"
