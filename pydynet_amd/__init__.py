"""pydynet_amd -- MI355X-native compute backend behind PyDyNet's Tensor / nn / autograd surface."""
