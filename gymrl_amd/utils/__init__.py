"""MI355X counterparts of the reference's shared helpers utils/{normalization,buffer,runner}.py."""
