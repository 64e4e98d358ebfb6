"""imageprocessing_amd -- MI355X-native hot path of fannz00/ImageProcessing."""
