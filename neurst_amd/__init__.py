"""MI355X-native SpeechTransformer training path behind NeurST's registries.

Importing the package changes nothing in the host process.  The one process-wide HIP setting the TRAINING path wants
(GPU_MAX_HW_QUEUES, see runtime.configure_training_process) is applied by the training entry points -- init_distributed(),
bench.py, the neurst-run CLI -- before they touch the device, and logged there.
"""
