"""raytracing_amd — MI355X-native drop-in for Takenoha/RayTracing's `Scene::simulate_rays` hot path."""
